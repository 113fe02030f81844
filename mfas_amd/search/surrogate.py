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
        per_cell = [self.embedding(cell) for cell in sequence_of_operations]
        hidden_states, _ = self.lstm(torch.stack(per_cell, dim=0))
        return self.nonlinearity(self.hid2val(hidden_states[-1]))

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


def train_simple_surrogate(model, criterion, optimizer, data_tensors, num_epochs, device):
    """surrogate.py:133-157: ``num_epochs`` passes over the buckets, one optimizer step per bucket; returns the last loss."""
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
