"""``BatchLenInfo`` with the API of the reference's ``punica/utils/cat_tensor.py:26-66``: how a flattened token batch
splits into prefill requests (each several tokens) followed by decode requests (one token each)."""
from __future__ import annotations

from typing import Sequence

import torch


class BatchLenInfo:
    def __init__(self, prefills: Sequence[int], decode: int, indptr_device: torch.device,
                 indptr_dtype: torch.dtype = torch.int32):
        self._prefills = list(prefills)
        self._decode = int(decode)
        self._doff = int(sum(self._prefills))
        self._indptr = None
        if self._prefills:
            ends = torch.tensor(self._prefills, dtype=torch.int64).cumsum(0)
            self._indptr = torch.cat([torch.zeros(1, dtype=torch.int64), ends]).to(dtype=indptr_dtype,
                                                                                   device=indptr_device)

    prefills = property(lambda self: self._prefills)      # length of each prefill request
    decode = property(lambda self: self._decode)          # number of decode requests
    doff = property(lambda self: self._doff)              # index of the first decode token = total prefill length
    indptr = property(lambda self: self._indptr)          # indptr[i] = sum(prefills[:i]); None without prefills
