"""Surrogate of the search: a small LSTM regressor conf -> accuracy, and its growing training set.

Semantics follow /root/reference/models/search/surrogate.py: SimpleRecurrentSurrogate :15-60 (Linear(3,100)+Sigmoid
-> LSTM(100,100) -> Linear(100,1) -> Sigmoid; Linear weights U(-0.1,0.1), biases 1.8), SurrogateDataloader :64-129
(dict per sequence length keyed by the conf bytes, keeps the max accuracy), train_simple_surrogate :133-157.
81,301 parameters: plain PyTorch on whatever device the caller passes (CPU is fine) — not part of the HIP hot path.
"""
import numpy as np
import torch
import torch.nn as nn


def _reference_init(module):
    """surrogate.py:26-30: every Linear starts at U(-0.1, 0.1) weights and bias 1.8 (the LSTM keeps PyTorch's default)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            m.weight.data.uniform_(-0.1, 0.1)
            m.bias.data.fill_(1.8)


class SimpleRecurrentSurrogate(nn.Module):
    """conf rows (s, v, nl) -> embedding -> LSTM over the cells -> predicted accuracy in (0, 1)."""

    def __init__(self, num_hidden=100, number_input_feats=3, size_ebedding=100):
        super().__init__()
        self.num_hidden = num_hidden
        self.embedding = nn.Sequential(nn.Linear(number_input_feats, size_ebedding), nn.Sigmoid())
        self.lstm = nn.LSTM(size_ebedding, num_hidden)
        self.hid2val = nn.Linear(num_hidden, 1)
        self.nonlinearity = nn.Sigmoid()
        _reference_init(self)

    def forward(self, sequence_of_operations):
        """(seq_len, batch, 3) float -> (batch, 1).  The embedding is applied cell by cell (one GEMM per position, as the
        reference does) so that batched and single-sequence calls go through the same kernels per position."""
        if sequence_of_operations.is_cuda:      # on a HIP device: plain GEMMs + elementwise ops, capturable in a graph
            return self.forward_unrolled(sequence_of_operations)
        per_cell = [self.embedding(cell) for cell in sequence_of_operations]
        hidden_states, _ = self.lstm(torch.stack(per_cell, dim=0))
        return self.nonlinearity(self.hid2val(hidden_states[-1]))

    def forward_unrolled(self, x):
        """The same network with the LSTM cell written out (gates i, f, g, o of nn.LSTM's packed weights, h0 = c0 = 0): no library
        RNN call, so one train step is a fixed sequence of GEMM / elementwise kernels that a HIP graph can replay."""
        lin = self.embedding[0]
        n, H = x.shape[1], self.num_hidden
        h = x.new_zeros(n, H)
        c = x.new_zeros(n, H)
        w_ih, w_hh, b_ih, b_hh = self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0
        for t in range(x.shape[0]):
            e = torch.sigmoid(torch.nn.functional.linear(x[t], lin.weight, lin.bias))
            gates = torch.nn.functional.linear(e, w_ih, b_ih) + torch.nn.functional.linear(h, w_hh, b_hh)
            i, f, g, o = gates.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
        return torch.sigmoid(torch.nn.functional.linear(h, self.hid2val.weight, self.hid2val.bias))

    def eval_model(self, sequence_of_operations_np, device):
        """One configuration (L, 3) -> python-indexable scalar (kept for API parity; the controller predicts in batches,
        tools.predict_accuracies_with_surrogate)."""
        seq = torch.from_numpy(np.expand_dims(sequence_of_operations_np, 1)).float().to(device)
        return self.forward(seq).cpu().data.numpy()[0, 0]


class SurrogateDataloader:
    """The surrogate's growing training set: one bucket per configuration length, one entry per distinct configuration
    (keyed by its bytes) holding the best accuracy seen for it; buckets and entries keep insertion order."""

    def __init__(self):
        self._dict_data = {}

    def add_datum(self, datum_conf, datum_acc):
        bucket = self._dict_data.setdefault(len(datum_conf), {})
        key = datum_conf.data.tobytes()
        seen = bucket.get(key)
        bucket[key] = (datum_conf, datum_acc if seen is None else max(datum_acc, seen[1]))

    def _entries(self):
        for bucket in self._dict_data.values():
            yield list(bucket.values())

    def get_data(self, to_torch=False):
        """([ (seq_len, n, 3) float32 per bucket ], [ (n, 1) float32 per bucket ])."""
        wrap = torch.from_numpy if to_torch else (lambda x: x)
        confs, accs = [], []
        for entries in self._entries():
            stacked = np.asarray([c for c, _ in entries], np.float32)            # (n, seq_len, 3)
            confs.append(wrap(np.ascontiguousarray(stacked.transpose(1, 0, 2))))
            accs.append(wrap(np.asarray([[a] for _, a in entries], np.float32)))
        return confs, accs

    def get_k_best(self, k):
        flat = [e for entries in self._entries() for e in entries]
        accs = np.array([a for _, a in flat])
        top = np.argpartition(accs, -k)[-k:]
        return [flat[i][0] for i in top], [accs[i] for i in top], top


class GraphedSurrogateTrainer:
    """train_simple_surrogate on a HIP device as REPLAYED GRAPHS.  One train step of the 81 k-parameter surrogate is ~150 tiny
    kernels; issued eagerly they are launch-bound (2.9 ms per step on MI355X, slower than the 4-thread CPU path's 1.9 ms), replayed
    as one graph 0.87 ms.  A graph has fixed shapes, so every bucket (configurations of one length) lives in a padded buffer of
    `capacity` rows with a 0/1 mask; the loss is sum(mask * err^2) / n == MSELoss over the n real rows.  One graph per length,
    captured at first use (the warm-up steps capture needs are undone: parameters and optimizer state are restored).
    Same semantics as the eager loop (num_epochs passes, one Adam step per bucket, returns the last loss); numerics are those of
    the device's GEMMs, so decisions may differ from the CPU path in the last digits — the reference pins (G8 / G9) test the CPU path."""

    def __init__(self, model, optimizer, capacity=2048):
        self.model, self.opt, self.cap = model, optimizer, int(capacity)
        self.dev = next(model.parameters()).device
        self.graphs = {}

    def _bucket(self, L):
        if L in self.graphs:
            return self.graphs[L]
        dev, cap = self.dev, self.cap
        b = {"x": torch.zeros(L, cap, 3, device=dev), "y": torch.zeros(cap, 1, device=dev), "mask": torch.zeros(cap, 1, device=dev),
             "inv_n": torch.zeros((), device=dev), "loss": torch.zeros((), device=dev)}

        def step():
            self.opt.zero_grad(set_to_none=False)
            err = self.model.forward_unrolled(b["x"]) - b["y"]
            loss = (b["mask"] * err * err).sum() * b["inv_n"]
            loss.backward()
            self.opt.step()
            return loss

        # warm-up on a side stream (lazy library / optimizer-state initialisation must not happen inside the capture), then put
        # parameters and optimizer state back: with the all-zero mask the gradients are zero, but Adam's step counter moved
        params = [q for q in self.model.parameters()]
        for q in params:
            if q.grad is None:
                q.grad = torch.zeros_like(q)
        saved_p = [q.detach().clone() for q in params]
        saved_o = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in self.opt.state.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            b["loss_live"] = step()
        with torch.no_grad():
            for q, v in zip(params, saved_p):
                q.copy_(v)
            for k, st in self.opt.state.items():
                for kk, vv in st.items():
                    if torch.is_tensor(vv):
                        if k in saved_o and kk in saved_o[k]:
                            vv.copy_(saved_o[k][kk])
                        else:
                            vv.zero_()       # state created by the warm-up: back to "never stepped"
        b["graph"] = g
        self.graphs[L] = b
        return b

    def train(self, data_tensors, num_epochs):
        buckets = []
        for x, y in zip(*data_tensors):
            L, n = int(x.shape[0]), int(x.shape[1])
            if n > self.cap:          # the training set outgrew the padded buffers: larger ones, graphs captured again
                while self.cap < n:
                    self.cap *= 2
                self.graphs.clear()
            b = self._bucket(L)
            b["x"].zero_(); b["y"].zero_(); b["mask"].zero_()
            b["x"][:, :n].copy_(x.to(self.dev)); b["y"][:n].copy_(y.to(self.dev)); b["mask"][:n].fill_(1.0)
            b["inv_n"].fill_(1.0 / n)
            buckets.append(b)
        self.model.train(True)
        last = None
        for _ in range(num_epochs):
            for b in buckets:
                b["graph"].replay()
                last = b
        self.model.train(False)
        return float(last["loss_live"].item())


def train_simple_surrogate(model, criterion, optimizer, data_tensors, num_epochs, device):
    """surrogate.py:133-157: ``num_epochs`` passes over the buckets, one optimizer step per bucket; returns the last loss."""
    if torch.device(device).type == "cuda" and isinstance(criterion, nn.MSELoss):
        tr = getattr(model, "_graphed_trainer", None)
        if tr is None or tr.opt is not optimizer:
            tr = GraphedSurrogateTrainer(model, optimizer)
            model._graphed_trainer = tr
        return tr.train(data_tensors, num_epochs)
    buckets = [(x.to(device), y.to(device)) for x, y in zip(*data_tensors)]
    last = None
    model.train(True)
    for _ in range(num_epochs):
        for x, y in buckets:
            optimizer.zero_grad()
            last = criterion(model(x), y)
            last.backward()
            optimizer.step()
    model.train(False)
    return last.item()
