"""Helpers to read the golden fixtures written by oracle/gen_golden.py."""
import json
import os

import numpy as np
import torch

from tests.conftest import GOLDEN_DIR


class Golden:
    def __init__(self, name):
        self.d = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(str(self.d['meta']))
        self.ncalls = self.meta['ncalls']

    def params(self, prefix='param/'):
        return {k[len(prefix):]: torch.from_numpy(self.d[k].copy()) for k in self.d.files if k.startswith(prefix)}

    def grads(self):
        return self.params('grad/')

    def final_buffers(self):
        return self.params('final/')

    def t(self, key):
        return torch.from_numpy(np.asarray(self.d[key]).copy())

    def has(self, key):
        return key in self.d.files

    def adjacency(self, c, which, device='cpu'):
        """Rebuild the adjacency exactly as the reference produced it (dense on the first
        CPU call, otherwise a sparse COO tensor with its original, possibly uncoalesced, entries)."""
        pre = f'c{c}/{which}'
        N = int(self.d[f'c{c}/N'])
        idx = torch.from_numpy(self.d[pre + '_idx'].copy())
        val = torch.from_numpy(self.d[pre + '_val'].copy())
        if int(self.d[pre + '_dense']):
            a = torch.zeros(N, N)
            a[idx[0], idx[1]] = val
            return a.to(device)
        return torch.sparse_coo_tensor(idx, val, (N, N)).to(device)
