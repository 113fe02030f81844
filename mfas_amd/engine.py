"""Python face of the HIP engine: hyper-parameters, feature tables, and the lockstep population.

PyTorch is used for device memory, the current HIP stream and ``torch.distributed`` only; all
arithmetic of the path runs inside ``libmfas_hip.so`` (mfas_amd/csrc/mfas_hip.hip).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .scheduler import adam_step_scalars

S_SIZES = (128, 256, 1024, 512)    # /root/reference/models/search/ntu_searchable.py:291
V_SIZES = (512, 1024, 2048, 2048)  # ntu_searchable.py:292
TAPS = tuple(f"s{j}" for j in range(_lib.MAX_TAPS)) + tuple(f"v{j}" for j in range(_lib.MAX_TAPS))


@dataclass
class Hyper:
    """The ``args`` fields the path reads (ntu_searchable.py:28-84,200-292)."""
    R: int = 16
    C: int = 60
    B: int = 20
    bn: bool = False
    drpt: float = 0.5
    alphas: bool = False
    multitask: bool = False
    wd: float = 1e-4            # ntu_searchable.py:65
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    s_sizes: Sequence[int] = S_SIZES
    v_sizes: Sequence[int] = V_SIZES
    loss_mode: int = 0          # 0 CE + top-1 (NTU); 1 weighted BCE-with-logits + F1-samples (MM-IMDB)
    f1_threshold: float = 0.3   # th_fscore, train_searchable/mmimdb.py:16
    allow_plain_cell: bool = False   # [Linear, nl] cells are legal (AV-MNIST, avmnist_searchable.py:276-285)
    tap_bits: int = 0           # element size of the feature tables to come (16 / 32; 0 = unknown): lets the engine size its units
    order_per_candidate: bool = False   # every candidate walks its own per-epoch permutations (order [K][E][N]) instead of sharing one

    @classmethod
    def from_args(cls, args) -> "Hyper":
        s_sizes = list(getattr(args, "s_sizes", S_SIZES))
        if not hasattr(args, "s_sizes") and hasattr(args, "vid_len"):
            s_sizes[2] = int(args.vid_len[1]) * 32      # ntu_searchable.py:291
        return cls(R=int(args.inner_representation_size), C=int(args.num_outputs), B=int(args.batchsize),
                   bn=bool(args.batchnorm), drpt=float(args.drpt), alphas=bool(args.alphas),
                   multitask=bool(getattr(args, "multitask", False)), s_sizes=tuple(s_sizes),
                   v_sizes=tuple(getattr(args, "v_sizes", V_SIZES)))

    def to_c(self) -> _lib.mfas_hyper:
        h = _lib.mfas_hyper()
        h.R, h.C, h.B = self.R, self.C, self.B
        h.bn, h.alphas, h.multitask = int(self.bn), int(self.alphas), int(self.multitask)
        h.drpt = float(self.drpt)
        h.wd, h.beta1, h.beta2 = self.wd, self.beta1, self.beta2
        h.adam_eps, h.bn_eps, h.bn_momentum = self.adam_eps, self.bn_eps, self.bn_momentum
        if len(self.s_sizes) > _lib.MAX_TAPS or len(self.v_sizes) > _lib.MAX_TAPS:
            raise ValueError(f"at most {_lib.MAX_TAPS} taps per modality")
        for j in range(_lib.MAX_TAPS):
            h.s_sizes[j] = int(self.s_sizes[j]) if j < len(self.s_sizes) else 0
            h.v_sizes[j] = int(self.v_sizes[j]) if j < len(self.v_sizes) else 0
        h.loss_mode = int(self.loss_mode)
        h.allow_plain_cell = int(self.allow_plain_cell)
        h.f1_threshold = float(self.f1_threshold)
        h.tap_bits = int(self.tap_bits)
        h.order_per_candidate = int(bool(self.order_per_candidate))
        return h


def cell_in_features(conf, i, hp: Hyper) -> int:
    return hp.s_sizes[int(conf[i][0])] + hp.v_sizes[int(conf[i][1])] + (hp.R if i > 0 else 0)


def flat_layout(conf, hp: Hyper):
    """[(state_dict key, shape, offset)] of a candidate's central parameters in the engine's flat
    order (= reference state_dict order, ntu_searchable.py:191-200), and the total float count."""
    out, off = [], 0
    L = len(conf)

    def put(key, shape):
        nonlocal off
        out.append((key, tuple(shape), off))
        off += int(np.prod(shape))

    for i in range(L):
        put(f"alphas.{i}.alpha_x", (1,))
    for i in range(L):
        K = cell_in_features(conf, i, hp)
        put(f"fusion_layers.{i}.0.weight", (hp.R, K))
        put(f"fusion_layers.{i}.0.bias", (hp.R,))
        if hp.bn:
            for nm in ("weight", "bias", "running_mean", "running_var"):
                put(f"fusion_layers.{i}.2.{nm}", (hp.R,))
    put("central_classifier.weight", (hp.C, hp.R))
    put("central_classifier.bias", (hp.C,))
    return out, off


# ------------------------------------------------------------------------------------------------
class FeatureTable:
    """Pooled backbone taps for N samples resident in HBM: what ``Visual``/``Skeleton`` +
    ``GlobalPooling2D`` hand to the fusion net (models/central/ntu.py:35-50,129-183;
    ntu_searchable.py:211-225) plus labels (datasets/ntu.py:254)."""

    def __init__(self, taps: Dict[str, torch.Tensor], label: torch.Tensor,
                 vlogit: Optional[torch.Tensor] = None, slogit: Optional[torch.Tensor] = None,
                 multilabel: Optional[torch.Tensor] = None):
        """label: (N,) class indices; for multi-label data pass multilabel (N, C) 0/1 targets as well (label may then
        be any (N,) tensor on the device, e.g. zeros)."""
        dev = label.device
        if dev.type != "cuda":
            raise RuntimeError("FeatureTable must live on a HIP device (cuda:N); there is no CPU path")
        dts = {t.dtype for t in taps.values()}
        if len(dts) != 1:
            raise ValueError("all taps must share one dtype")
        self.dtype = dts.pop()
        if self.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"unsupported tap dtype {self.dtype}")
        self.N = int(label.shape[0])
        self.widths = {k: int(v.shape[1]) for k, v in taps.items()}     # true tap widths
        self.taps = {}
        for k, v in taps.items():
            if v.shape[0] != self.N or v.dim() != 2 or v.device != dev:
                raise ValueError(f"tap {k}: expected (N, width) on {dev}")
            pad = (-v.shape[1]) % 16       # the engine reads rows padded with zeros to a multiple of 16 elements
            self.taps[k] = (torch.nn.functional.pad(v, (0, pad)) if pad else v).contiguous()
        self.label = label.to(torch.int32).contiguous()
        self.vlogit = None if vlogit is None else vlogit.to(torch.float32).contiguous()
        self.slogit = None if slogit is None else slogit.to(torch.float32).contiguous()
        self.multilabel = None if multilabel is None else multilabel.to(torch.float32).contiguous()
        self.device = dev

    def check_labels(self, C: int):
        """Labels must lie in [0, C) (checked once per table and class count; one small device reduction)."""
        if getattr(self, "_labels_ok", None) == C:
            return
        lo, hi = int(self.label.min()), int(self.label.max())
        if lo < 0 or hi >= C:
            raise IndexError(f"Target {hi if hi >= C else lo} is out of bounds for {C} classes")
        self._labels_ok = C

    def __len__(self):
        return self.N

    @classmethod
    def from_numpy(cls, table: Dict[str, np.ndarray], device, dtype=torch.float32) -> "FeatureTable":
        taps = {k: torch.from_numpy(np.ascontiguousarray(table[k])).to(device=device, dtype=dtype)
                for k in TAPS if k in table}
        lab = torch.from_numpy(np.asarray(table["label"]).astype(np.int32)).to(device)
        vl = torch.from_numpy(table["vlogit"]).to(device) if "vlogit" in table else None
        sl = torch.from_numpy(table["slogit"]).to(device) if "slogit" in table else None
        return cls(taps, lab, vl, sl)

    # ---- on-disk format: one .npy per tap; bf16 stored as its uint16 bit pattern
    def save(self, directory: str, split: str):
        import os
        os.makedirs(directory, exist_ok=True)
        for k, v in self.taps.items():
            a = v[:, :self.widths[k]].cpu()
            if a.dtype == torch.bfloat16:
                np.save(os.path.join(directory, f"{split}_{k}.bf16.npy"), a.view(torch.int16).numpy().view(np.uint16))
            else:
                np.save(os.path.join(directory, f"{split}_{k}.npy"), a.numpy())
        np.save(os.path.join(directory, f"{split}_label.npy"), self.label.cpu().numpy())
        if self.vlogit is not None:
            np.save(os.path.join(directory, f"{split}_vlogit.npy"), self.vlogit.cpu().numpy())
            np.save(os.path.join(directory, f"{split}_slogit.npy"), self.slogit.cpu().numpy())

    @classmethod
    def load(cls, directory: str, split: str, device) -> "FeatureTable":
        import os
        taps = {}
        for k in TAPS:
            p16 = os.path.join(directory, f"{split}_{k}.bf16.npy")
            p = os.path.join(directory, f"{split}_{k}.npy")
            if os.path.exists(p16):
                taps[k] = torch.from_numpy(np.load(p16).view(np.int16)).view(torch.bfloat16).to(device)
            elif os.path.exists(p):
                taps[k] = torch.from_numpy(np.load(p)).to(device)
        if not taps:
            raise FileNotFoundError(f"no '{split}_<tap>.npy' feature files under {directory}")
        lab = torch.from_numpy(np.load(os.path.join(directory, f"{split}_label.npy")).astype(np.int32)).to(device)
        vl = sl = None
        if os.path.exists(os.path.join(directory, f"{split}_vlogit.npy")):
            vl = torch.from_numpy(np.load(os.path.join(directory, f"{split}_vlogit.npy"))).to(device)
            sl = torch.from_numpy(np.load(os.path.join(directory, f"{split}_slogit.npy"))).to(device)
        return cls(taps, lab, vl, sl)

    @classmethod
    def synthetic(cls, N: int, seed: int, device, dtype=torch.bfloat16, snr=0.15, C=60, with_logits=False,
                  s_sizes=S_SIZES, v_sizes=V_SIZES, mu_seed=123) -> "FeatureTable":
        """Planted-signal NTU-shaped taps x = relu(snr*mu[label] + eps) generated on the device."""
        g = torch.Generator(device=device)
        g.manual_seed(mu_seed)
        mus = {}
        for name, sizes in (("s", s_sizes), ("v", v_sizes)):
            for j, w in enumerate(sizes):
                mus[f"{name}{j}"] = torch.randn(C, w, generator=g, device=device)
        mul = [torch.randn(C, C, generator=g, device=device) for _ in range(2)]
        g.manual_seed(seed)
        label = torch.randint(0, C, (N,), generator=g, device=device)
        taps = {k: torch.relu(snr * mu[label] + torch.randn(N, mu.shape[1], generator=g, device=device)).to(dtype)
                for k, mu in mus.items()}
        vl = sl = None
        if with_logits:
            vl, sl = [0.5 * m[label] + torch.randn(N, C, generator=g, device=device) for m in mul]
        return cls(taps, label.to(torch.int32), vl, sl)

    def to_c(self) -> _lib.mfas_table:
        t = _lib.mfas_table()
        some = next(iter(self.taps.values())).data_ptr()
        for j in range(_lib.MAX_TAPS):   # taps a population never selects may be absent: any valid pointer will do
            t.s[j] = self.taps[f"s{j}"].data_ptr() if f"s{j}" in self.taps else some
            t.v[j] = self.taps[f"v{j}"].data_ptr() if f"v{j}" in self.taps else some
        t.vlogit = None if self.vlogit is None else self.vlogit.data_ptr()
        t.slogit = None if self.slogit is None else self.slogit.data_ptr()
        t.label = self.label.data_ptr()
        t.multilabel = None if self.multilabel is None else self.multilabel.data_ptr()
        t.N = self.N
        t.dtype = _lib.MFAS_DT[str(self.dtype).replace("torch.", "")]
        return t

    def elem_size(self) -> int:
        return 4 if self.dtype == torch.float32 else 2


class FeatureLoader:
    """What ``dataloaders['train'|'dev'|'test']`` is for the engine: the reference iterates a
    DataLoader of {'rgb','ske','label'} batches (train_searchable/ntu.py:35-43); here the whole
    table already sits in HBM and only the batch order remains of the loader."""

    def __init__(self, table: FeatureTable, batch_size: int, shuffle: bool = True):
        self.table = table
        self.dataset = table            # len(loader.dataset) is read at ntu_searchable.py:29
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)

    def __len__(self):
        return -(-len(self.table) // self.batch_size)


# ------------------------------------------------------------------------------------------------
class Population:
    """K candidates trained in lockstep on one GPU (handle on ``mfas_population``)."""

    def __init__(self, hp: Hyper, confs: Sequence[np.ndarray], device, drop_seeds=None, chunk_cols=0):
        self.lib = _lib.lib()
        self.hp = hp
        self.confs = [np.asarray(c, dtype=np.int64).reshape(-1, 3) for c in confs]
        self.K = len(self.confs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the MFAS engine runs on a HIP device only (no CPU fallback)")
        cf = np.zeros((self.K, 4, 3), np.int32)
        nc = np.zeros(self.K, np.int32)
        for k, c in enumerate(self.confs):
            if not 1 <= len(c) <= 4:
                raise ValueError("a configuration has 1..4 fusion cells")
            cf[k, :len(c)] = c
            nc[k] = len(c)
        ds = None if drop_seeds is None else np.asarray(drop_seeds, np.uint32)
        self._used_taps = {}        # tap name -> (declared width, first configuration / cell that selects it)
        for k, c in enumerate(self.confs):
            for i, (sj, vj, _) in enumerate(c):
                if 0 <= sj < len(hp.s_sizes):
                    self._used_taps.setdefault(f"s{int(sj)}", (int(hp.s_sizes[int(sj)]), k, i))
                if 0 <= vj < len(hp.v_sizes):
                    self._used_taps.setdefault(f"v{int(vj)}", (int(hp.v_sizes[int(vj)]), k, i))
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            stream = torch.cuda.current_stream().cuda_stream
            hc = hp.to_c()
            _lib.check(self.lib.mfas_population_create(
                C.byref(hc), cf.ctypes.data, nc.ctypes.data, None if ds is None else ds.ctypes.data,
                self.K, idx, C.c_void_p(stream), int(chunk_cols), C.byref(self._h)))
        self._idx = idx

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.mfas_population_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_table(self, table: "FeatureTable"):
        """Every tap a configuration of this population selects must be present in the table with exactly the width the
        hyper-parameters declare (the kernels index rows with stride ceil16(width)); the reference raises a Linear shape
        error in that case (ntu_searchable.py:236-240)."""
        for name, (want, k, i) in self._used_taps.items():
            have = table.widths.get(name)
            if have is None:
                raise ValueError(f"configuration {k} cell {i} selects tap {name}, which the feature table does not hold")
            if have != want:
                raise ValueError(f"tap {name}: the feature table is {have} wide but the hyper-parameters declare {want} "
                                 "(mat1 and mat2 shapes cannot be multiplied in the reference)")

    def param_count(self, k: int) -> int:
        n = self.lib.mfas_population_param_count(self._h, k)
        if n < 0:
            _lib.check(int(n))
        return int(n)

    def set_params(self, k: int, flat: torch.Tensor, sync: bool = True):
        flat = flat.to(device=self.device, dtype=torch.float32).contiguous()
        assert flat.numel() == self.param_count(k), (flat.numel(), self.param_count(k))
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_set_params(self._h, k, C.c_void_p(flat.data_ptr())))
        if sync:   # flat may be freed by the caller (sync=False: the caller keeps it alive until it has synchronised the stream)
            torch.cuda.current_stream(self._idx).synchronize()

    def get_params(self, k: int, plane: int = 0) -> torch.Tensor:
        out = torch.empty(self.param_count(k), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mfas_population_get_params(self._h, k, plane, C.c_void_p(out.data_ptr())))
        return out

    def set_state_dict(self, k: int, sd: Dict[str, "np.ndarray | torch.Tensor"]):
        layout, n = flat_layout(self.confs[k], self.hp)
        flat = torch.zeros(n, dtype=torch.float32)
        for key, shape, off in layout:
            if key in sd:
                v = sd[key]
                v = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v.detach().cpu()
                flat[off:off + v.numel()] = v.reshape(-1).to(torch.float32)
            elif key.endswith(("2.weight", "running_var")):
                flat[off:off + int(np.prod(shape))] = 1.0
        self.set_params(k, flat)

    def get_state_dict(self, k: int, plane: int = 0) -> Dict[str, torch.Tensor]:
        layout, _ = flat_layout(self.confs[k], self.hp)
        flat = self.get_params(k, plane).cpu()
        return {key: flat[off:off + int(np.prod(shape))].reshape(shape).clone() for key, shape, off in layout}

    def init(self, seeds: Sequence[int]):
        s = np.asarray(seeds, np.uint32)
        assert len(s) == self.K
        _lib.check(self.lib.mfas_population_init(self._h, s.ctypes.data))

    def init_torch_streams(self, seeds: Sequence[int], bounds: np.ndarray, alpha_mean: float = 0.0, alpha_std: float = 0.1):
        """mfas_population_init_torch_streams: every candidate's construction draws under torch.manual_seed(seeds[k]), generated on
        the device.  bounds: float32 [K][2 * (MAX_CELLS + 1)] (per cell weight / bias bound, then the classifier's)."""
        s = np.ascontiguousarray(np.asarray(seeds, np.uint64))
        b = np.ascontiguousarray(np.asarray(bounds, np.float32))
        assert len(s) == self.K and b.shape == (self.K, 10), (len(s), b.shape)
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_init_torch_streams(self._h, s.ctypes.data, b.ctypes.data, float(alpha_mean), float(alpha_std)))

    def train(self, train: FeatureTable, dev: Optional[FeatureTable], epochs: int, etas: np.ndarray,
              order: Optional[torch.Tensor] = None, max_steps: int = -1, snapshot_best: bool = False):
        """Runs train_ntu_track_acc for the whole population.  Returns (stats, status): stats is a
        structured array [K, epochs] with train_loss_sum, dev_loss_sum, train_corrects, dev_corrects."""
        nb = -(-len(train) // self.hp.B)
        etas = np.asarray(etas, np.float64)
        if len(etas) < epochs * nb and max_steps < 0:
            raise ValueError("eta table shorter than epochs * batches")
        sc = np.ascontiguousarray(adam_step_scalars(etas, self.hp.beta1, self.hp.beta2))
        if order is not None:
            order = order.to(device=self.device, dtype=torch.int32).contiguous()
            assert order.numel() >= epochs * len(train) * (self.K if self.hp.order_per_candidate else 1), \
                "order: [epochs][N_train] (shared) or [K][epochs][N_train] (Hyper.order_per_candidate)"
        if self.hp.loss_mode == 0:      # CrossEntropyLoss raises on a target outside [0, C); the kernels index by it
            for t in (train, dev):
                if t is not None:
                    t.check_labels(self.hp.C)
        for t in (train, dev):
            if t is not None:
                self._check_table(t)
        stats = np.zeros((self.K, epochs), dtype=[("train_loss_sum", "f8"), ("dev_loss_sum", "f8"),
                                                  ("train_corrects", "i8"), ("dev_corrects", "i8")])
        status = np.zeros(self.K, np.int32)
        tt = train.to_c()
        td = dev.to_c() if dev is not None else None
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_train(
                self._h, C.byref(tt), C.byref(td) if td is not None else None,
                None if order is None else C.c_void_p(order.data_ptr()), sc.ctypes.data, int(epochs),
                int(max_steps), int(snapshot_best), stats.ctypes.data, status.ctypes.data))
        return stats, status

    def forward(self, k: int, table: FeatureTable, row0: int = 0, nrows: Optional[int] = None,
                count: bool = False):
        nrows = len(table) - row0 if nrows is None else nrows
        self._check_table(table)
        logits = torch.empty((nrows, self.hp.C), dtype=torch.float32, device=self.device)
        corr = C.c_int64(0)
        tc = table.to_c()
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_forward(self._h, k, C.byref(tc), row0, nrows,
                                                        C.c_void_p(logits.data_ptr()),
                                                        C.byref(corr) if count else None))
        return (logits, int(corr.value)) if count else logits

    def forward_train(self, k: int, table: FeatureTable, row0: int = 0, nrows: Optional[int] = None, step: int = 0):
        """Train-mode forward of ONE batch (<= hp.B rows): batch-statistics BN (running stats move), dropout stream at `step`."""
        nrows = len(table) - row0 if nrows is None else nrows
        self._check_table(table)
        logits = torch.empty((nrows, self.hp.C), dtype=torch.float32, device=self.device)
        tc = table.to_c()
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_forward_train(self._h, k, C.byref(tc), row0, nrows, int(step),
                                                              C.c_void_p(logits.data_ptr())))
        return logits

    def backward(self, k: int, table: FeatureTable, dlogits: torch.Tensor, row0: int = 0, nrows: Optional[int] = None, step: int = 0):
        """Gradients of an external loss of ONE train-mode batch (mfas_population_backward): `dlogits` = dL/dlogits (nrows, C) on
        the device.  Returns the flat gradient vector (reference state_dict order, like get_params); parameters are unchanged, the
        Adam slots and BN running statistics of this population are scratch afterwards."""
        nrows = len(table) - row0 if nrows is None else nrows
        self._check_table(table)
        d = dlogits.to(device=self.device, dtype=torch.float32).contiguous()
        assert tuple(d.shape) == (nrows, self.hp.C), (tuple(d.shape), (nrows, self.hp.C))
        tc = table.to_c()
        with torch.cuda.device(self._idx):
            _lib.check(self.lib.mfas_population_backward(self._h, k, C.byref(tc), row0, nrows, int(step), C.c_void_p(d.data_ptr())))
        return self.get_params(k, plane=1)

    def set_pos_weight(self, w):
        w = np.ascontiguousarray(np.asarray(w, np.float32))
        assert w.size == self.hp.C
        _lib.check(self.lib.mfas_population_set_pos_weight(self._h, w.ctypes.data))

    def set_best_threshold(self, threshold: float):
        """snapshot_best: the dev metric an epoch must EXCEED to replace the kept parameters (init_f1; 0 for accuracy)."""
        _lib.check(self.lib.mfas_population_set_best_threshold(self._h, float(threshold)))

    def set_profiling(self, on: bool):
        _lib.check(self.lib.mfas_population_set_profiling(self._h, int(on)))

    def schedule(self):
        """The step schedule the engine laid this population out for (mfas_population_schedule)."""
        info = (C.c_int32 * 8)()
        _lib.check(self.lib.mfas_population_schedule(self._h, info))
        keys = ("persistent", "resident_units", "resident_workgroups", "units_per_workgroup", "resident_chain", "lean_chain", "groups", "candidates")
        d = dict(zip(keys, (int(x) for x in info)))
        # info[5] bits 8..15: CUs one candidate's general chain runs on in the same-group launch (round 6: chain_split; 1 otherwise)
        d["chain_cus"] = max(1, (d["lean_chain"] >> 8) & 0xFF)
        d["lean_chain"] &= 0xFF
        return d

    def sweep_profile(self):
        n, ms, by = C.c_int64(0), C.c_double(0), C.c_double(0)
        _lib.check(self.lib.mfas_population_sweep_profile(self._h, C.byref(n), C.byref(ms), C.byref(by)))
        return int(n.value), float(ms.value), float(by.value)


F1_FIXED_POINT = float(1 << 32)


def plan_population(hp: Hyper, confs: Sequence[np.ndarray], device, chunk_cols: int = 0) -> Dict[str, int]:
    """mfas_population_plan: the schedule mfas_population_create would lay these configurations out for — a pure query (nothing
    is allocated or launched).  Keys as Population.schedule() where they apply."""
    lib = _lib.lib()
    confs = [np.asarray(c, dtype=np.int64).reshape(-1, 3) for c in confs]
    cf = np.zeros((len(confs), 4, 3), np.int32)
    nc = np.zeros(len(confs), np.int32)
    for k, c in enumerate(confs):
        if not 1 <= len(c) <= 4:
            raise ValueError("a configuration has 1..4 fusion cells")
        cf[k, :len(c)] = c
        nc[k] = len(c)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    info = np.zeros(8, np.int32)
    hc = hp.to_c()
    _lib.check(lib.mfas_population_plan(C.byref(hc), cf.ctypes.data, nc.ctypes.data, len(confs), int(idx), int(chunk_cols), info.ctypes.data))
    return {"persistent": bool(info[0]), "resident_units": int(info[1]), "resident_workgroups": int(info[2]),
            "units_per_workgroup": int(info[3]), "chunk_cols": int(info[4]), "lean_chain": bool(info[5]), "compute_units": int(info[6]),
            "candidates": int(info[7])}


def best_dev_f1(stats_row, status_nan: bool, n_dev: int, init_f1: float = 0.0):
    """train_mmimdb_track_f1's bookkeeping (train_searchable/mmimdb.py:18-137): best F1-samples over the epochs,
    strict '>' from init_f1; a NaN train-epoch loss ends the run with the best so far; a NaN best becomes 0."""
    best = init_f1
    for e in range(len(stats_row)):
        if status_nan and not np.isfinite(stats_row["train_loss_sum"][e]):
            break
        f1 = float(stats_row["dev_corrects"][e]) / F1_FIXED_POINT / float(n_dev)
        if f1 > best:
            best = f1
    return 0.0 if best != best else best


def best_dev_accuracy(stats_row, n_dev: int) -> float:
    """max over epochs with strict '>' from 0 (train_searchable/ntu.py:18,82-83); float64 ratio (:76)."""
    best = 0.0
    for e in range(len(stats_row)):
        acc = float(stats_row["dev_corrects"][e]) / float(n_dev)
        if acc > best:
            best = acc
    return best
