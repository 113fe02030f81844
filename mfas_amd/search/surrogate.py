"""Surrogate of the search: a small LSTM regressor conf -> accuracy, and its growing training set.

Semantics follow /root/reference/models/search/surrogate.py: SimpleRecurrentSurrogate :15-60 (Linear(3,100)+Sigmoid
-> LSTM(100,100) -> Linear(100,1) -> Sigmoid; Linear weights U(-0.1,0.1), biases 1.8), SurrogateDataloader :64-129
(dict per sequence length keyed by the conf bytes, keeps the max accuracy), train_simple_surrogate :133-157.
81,301 parameters: plain PyTorch on whatever device the caller passes (CPU is fine) — not part of the HIP hot path.
"""
import numpy as np
import torch
import torch.nn as nn


class SimpleRecurrentSurrogate(nn.Module):
    def __init__(self, num_hidden=100, number_input_feats=3, size_ebedding=100):
        super().__init__()
        self.num_hidden = num_hidden
        self.embedding = nn.Sequential(nn.Linear(number_input_feats, size_ebedding), nn.Sigmoid())
        self.lstm = nn.LSTM(size_ebedding, num_hidden)
        self.hid2val = nn.Linear(num_hidden, 1)
        self.nonlinearity = nn.Sigmoid()
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.uniform_(-0.1, 0.1)
                m.bias.data.fill_(1.8)

    def forward(self, sequence_of_operations):
        """(seq_len, batch, 3) -> (batch, 1)."""
        embeds = torch.stack([self.embedding(s) for s in sequence_of_operations], dim=0)
        lstm_out, _ = self.lstm(embeds)
        return self.nonlinearity(self.hid2val(lstm_out[-1]))

    def eval_model(self, sequence_of_operations_np, device):
        seq = torch.from_numpy(np.expand_dims(sequence_of_operations_np, 1)).float().to(device)
        return self.forward(seq).cpu().data.numpy()[0, 0]


class SurrogateDataloader:
    def __init__(self):
        self._dict_data = {}

    def add_datum(self, datum_conf, datum_acc):
        bucket = self._dict_data.setdefault(len(datum_conf), {})
        key = datum_conf.data.tobytes()
        if key in bucket:
            datum_acc = max(datum_acc, bucket[key][1])      # keep the best accuracy seen for a conf
        bucket[key] = (datum_conf, datum_acc)

    def get_data(self, to_torch=False):
        confs, accs = [], []
        for _, bucket in self._dict_data.items():
            c = np.asarray([d[0] for d in bucket.values()], np.float32)
            confs.append(np.array(np.transpose(c, (1, 0, 2)), np.float32))          # (seq_len, n, 3)
            accs.append(np.expand_dims(np.array([d[1] for d in bucket.values()], np.float32), 1))
        if to_torch:
            confs = [torch.from_numpy(c) for c in confs]
            accs = [torch.from_numpy(a) for a in accs]
        return confs, accs

    def get_k_best(self, k):
        confs, accs = [], []
        for _, bucket in self._dict_data.items():
            for d in bucket.values():
                confs.append(d[0])
                accs.append(d[1])
        accs = np.array(accs)
        top = np.argpartition(accs, -k)[-k:]
        return [confs[i] for i in top], [accs[i] for i in top], top


def train_simple_surrogate(model, criterion, optimizer, data_tensors, num_epochs, device):
    loss = None
    for _ in range(num_epochs):
        model.train(True)
        for inputs, outputs in zip(data_tensors[0], data_tensors[1]):
            inputs, outputs = inputs.to(device), outputs.to(device)
            optimizer.zero_grad()
            with torch.set_grad_enabled(True):
                loss = criterion(model(inputs), outputs)
                loss.backward()
                optimizer.step()
    model.train(False)
    return loss.item()
