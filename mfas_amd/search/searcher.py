"""Search drivers: own, importable counterpart of /root/reference/models/searchable.py (which cannot be imported as
shipped: it pulls `models.aux.scheduler`).

ModelSearcher._epnas :48-137 (sequential model-based search with an LSTM surrogate and temperature sampling),
._randsearch :139-174, NTUSearcher :233-260.  The inner candidate training is the HIP engine
(mfas_amd.train_sampled_models); data are HBM-resident feature tables instead of raw-video DataLoaders.
"""
import numpy as np
import torch
import torch.optim as op

from .. import avmnist_searchable as avmnist
from .. import mmimdb_searchable as mmimdb
from .. import ntu_searchable as ntu
from ..engine import FeatureLoader
from . import surrogate as surr
from . import tools


class ModelSearcher:
    def __init__(self, args):
        self.args = args

    def search(self):
        pass

    def _epnas(self, model_type, surrogate_dict, dataloaders, dataset_searchmethods, device):
        a = self.args
        surrogate, s_crite = surrogate_dict["model"], surrogate_dict["criterion"]
        s_data = surr.SurrogateDataloader()
        on_gpu = torch.device(device).type == "cuda"      # (graph-replayed train steps need the capturable optimizer)
        s_optim = op.Adam(surrogate.parameters(), lr=a.lr_surrogate, **({"capturable": True, "foreach": True} if on_gpu else {}))
        train_sampled_models = dataset_searchmethods["train_sampled_fun"]
        get_layer_confs = dataset_searchmethods["get_layer_confs"]
        temperature = a.initial_temperature
        sampled_k_confs = []
        shared_weights = dict()
        for si in range(a.search_iterations):
            if a.verbose:
                print(50 * "=")
                print("Search iteration {}/{} ".format(si, a.search_iterations))
            for pi in range(a.max_progression_levels):
                if a.verbose:
                    print(25 * "-")
                    print("Progressive step {}/{} ".format(pi, a.max_progression_levels))
                all_confs = tools.merge_unfolded_with_sampled(sampled_k_confs, get_layer_confs(pi), pi)
                first = si + pi == 0
                if first:   # very first step: every single-layer conf is really trained (:87-93)
                    all_accs = train_sampled_models(all_confs, model_type, dataloaders, a, device,
                                                    state_dict=shared_weights)
                    tools.update_surrogate_dataloader(s_data, all_confs, all_accs)
                    tools.train_surrogate(surrogate, s_data, s_optim, s_crite, a, device)
                    if a.verbose:
                        print("Trained architectures: ")
                        print(list(zip(all_confs, all_accs)))
                else:       # afterwards the surrogate ranks the unfolded candidates (:98-102)
                    all_accs = tools.predict_accuracies_with_surrogate(all_confs, surrogate, device)
                    if a.verbose:
                        print("Predicted accuracies: ")
                        print(list(zip(all_confs, all_accs)))
                sampled_k_confs = tools.sample_k_configurations(all_confs, all_accs, a.num_samples, temperature)
                if first:
                    if a.verbose:
                        est = tools.predict_accuracies_with_surrogate(all_confs, surrogate, device)
                        print("Error on accuracies = {}".format(np.abs(np.array(est) - np.array(all_accs))))
                else:       # the K sampled ones are trained for real and fed back to the surrogate (:118-124)
                    sampled_k_accs = train_sampled_models(sampled_k_confs, model_type, dataloaders, a, device,
                                                          state_dict=shared_weights)
                    tools.update_surrogate_dataloader(s_data, sampled_k_confs, sampled_k_accs)
                    err = tools.train_surrogate(surrogate, s_data, s_optim, s_crite, a, device)
                    if a.verbose:
                        print("Trained architectures: ")
                        print(list(zip(sampled_k_confs, sampled_k_accs)))
                        print("with surrogate error: {}".format(err))
                # NB the reference indexes the schedule with si*search_iterations (not *max_progression_levels), :132
                temperature = tools.compute_temperature(si * a.search_iterations + pi, a)
                if a.verbose:
                    print("Temperature is being set to {}".format(temperature))
        return s_data

    def _randsearch(self, model_type, dataloaders, dataset_searchmethods, device):
        a = self.args
        s_data = surr.SurrogateDataloader()
        train_sampled_models = dataset_searchmethods["train_sampled_fun"]
        get_layer_confs = dataset_searchmethods["get_layer_confs"]
        shared_weights = dict()
        for si in range(a.search_iterations * a.max_progression_levels):
            if a.verbose:
                print(50 * "=")
                print("Random Search iteration {}/{} ".format(si, a.search_iterations * a.max_progression_levels))
            confs = tools.sample_k_configurations_directly(a.num_samples, a.max_progression_levels, get_layer_confs)
            accs = train_sampled_models(confs, model_type, dataloaders, a, device, state_dict=shared_weights)
            tools.update_surrogate_dataloader(s_data, confs, accs)
            if a.verbose:
                print("Trained architectures: ")
                print(list(zip(confs, accs)))
        return s_data


class _TableSearcher(ModelSearcher):
    """A searcher over HIP-resident feature tables: `tables` = {'train': FeatureTable, 'dev': FeatureTable}.  Subclasses
    name the searchable module (`model_type`) and the module providing train_sampled_models /
    get_possible_layer_configurations (`methods_module`)."""
    model_type = None
    methods_module = None
    shuffle = {"train": True, "dev": True}

    def __init__(self, args, device, tables):
        super().__init__(args)
        self.device = device
        self.dataloaders = {x: FeatureLoader(tables[x], args.batchsize, shuffle=self.shuffle[x]) for x in ("train", "dev")}

    def _methods(self):
        return {"train_sampled_fun": self.methods_module.train_sampled_models,
                "get_layer_confs": self.methods_module.get_possible_layer_configurations}

    def search(self, surrogate_device="cpu"):
        methods = self._methods()
        if getattr(self.args, "randsearch", False):
            return self._randsearch(self.model_type, self.dataloaders, methods, self.device)
        surrogate = surr.SimpleRecurrentSurrogate(100, 3, 100).to(surrogate_device)
        surrogate_dict = {"model": surrogate, "criterion": torch.nn.MSELoss()}
        # the (81k-parameter) surrogate lives where the caller wants it; the candidates train on self.device
        if str(surrogate_device) != str(self.device):
            return self._epnas_split(surrogate_dict, methods, self.device, surrogate_device)
        return self._epnas(self.model_type, surrogate_dict, self.dataloaders, methods, self.device)

    def _epnas_split(self, surrogate_dict, methods, train_device, surrogate_device):
        """_epnas with the candidates on `train_device` and the (tiny) surrogate on `surrogate_device`."""
        inner = methods["train_sampled_fun"]

        def train_on_gpu(confs, model_type, dataloaders, args, device, **kw):
            return inner(confs, model_type, dataloaders, args, train_device, **kw)

        m = dict(methods, train_sampled_fun=train_on_gpu)
        return self._epnas(self.model_type, surrogate_dict, self.dataloaders, m, surrogate_device)


class NTUSearcher(_TableSearcher):
    """models/searchable.py:233-260 on feature tables ('trainexp' / 'dev' splits); both loaders shuffle like the
    reference's DataLoaders (:248)."""
    model_type = ntu.Searchable_Skeleton_Image_Net
    methods_module = ntu


class AVMNISTSearcher(_TableSearcher):
    """models/searchable.py:184-227 on feature tables (train = samples 0..49,999 shuffled, dev = 50,000..54,999 in
    order); `args.randsearch` selects the random-search driver as in the reference."""
    shuffle = {"train": True, "dev": False}
    model_type = avmnist.Searchable_Audio_Image_Net
    methods_module = avmnist


class MMIMDBSearcher(_TableSearcher):
    """The reference has no MM-IMDB searcher; this is the NTU one bound to the MM-IMDB searchable."""
    shuffle = {"train": True, "dev": False}
    model_type = mmimdb.Searchable_Text_Image_Net
    methods_module = mmimdb


def timed_search(searcher, seed=0, surrogate_device="cpu"):
    """Run `searcher.search()` under fixed seeds (torch / numpy / random: every decision of the controller is then a function of
    the accuracies the engine returns) with the wall time split into candidate training and controller / surrogate, and a digest
    of the decision stream (every configuration list the controller asked to be trained, in order).  Returns
    (surrogate dataset, report dict).  Used by bench.py's `config.search_c3` entry (BASELINE
    configs[3]: --num_samples 50 --search_iterations 5 --max_fusions 4 -> 20 calls, 982 candidates) and by the GPU test of that schedule.
    (`main_searchable_ntu.py --timing` keeps its own, smaller wrapper: it must time the USER's run — unseeded, random search and N > 1
    ranks included — without reseeding it or hashing its decisions.)"""
    import hashlib
    import random
    import time
    methods = searcher._methods()
    inner = methods["train_sampled_fun"]
    spent = {"train_s": 0.0, "calls": 0, "candidates": 0, "call_sizes": []}
    h = hashlib.sha256()
    first = hashlib.sha256()

    def timed(confs, *a, **kw):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t = time.perf_counter()
        out = inner(confs, *a, **kw)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        spent["train_s"] += time.perf_counter() - t
        blob = b"".join(np.ascontiguousarray(np.asarray(c, np.int64)).tobytes() + b"|" for c in confs)
        if spent["calls"] == 0:
            first.update(blob)
        h.update(blob)
        spent["calls"] += 1
        spent["candidates"] += len(confs)
        spent["call_sizes"].append(len(confs))
        return out

    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)                # tools.sample_k_configurations_directly draws depths with random.randint
    searcher._methods = lambda: dict(methods, train_sampled_fun=timed)
    try:
        t0 = time.perf_counter()
        data = searcher.search(surrogate_device=surrogate_device)
        total = time.perf_counter() - t0
    finally:
        del searcher._methods        # (the instance attribute shadowing the class method)
    k_best, k_accs, _ = data.get_k_best(1)
    rep = {"candidates": spent["candidates"], "calls": spent["calls"], "call_sizes": spent["call_sizes"], "total_s": total,
           "train_s": spent["train_s"], "controller_s": total - spent["train_s"],
           "cand_per_s": spent["candidates"] / max(spent["train_s"], 1e-9), "cand_per_s_end_to_end": spent["candidates"] / max(total, 1e-9),
           "best_dev_acc": float(k_accs[0]) if len(k_accs) else None, "seed": seed,
           "first_call_digest": first.hexdigest()[:16], "decision_digest": h.hexdigest()[:16]}
    return data, rep
