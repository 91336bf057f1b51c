"""Memory-mapped token store and index builders with the reference's Python surface, backed by libdb1_data.so
(include/db1_data.h; C++ like the reference's pybind11 ``helpers`` module).

* ``MMapIndexedDataset(path_prefix)``: ``len()``, ``ds[i]``, ``ds[a:b]``, ``get(i, offset, length)``, ``sizes``, ``doc_idx``
  (src/data/indexed_dataset.py:351-563).  Items are zero-copy NumPy views into the mapping.
* ``build_sample_idx``, ``build_rl_sample_idx``, ``build_blending_indices``: same arguments and results as ``helpers.*``
  (src/data/helpers.cpp:20-203; callers gpt_dataset.py:287, rl_dataset.py:275, blendable_dataset.py:100).
"""
from __future__ import annotations

import ctypes
import os
from itertools import accumulate

import numpy as np

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_HERE, "libdb1_data.so")
_lib = None

# dtype codes of the .idx header (indexed_dataset.py:101-112; 6 is numpy's removed alias np.float = float64)
DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float64, 8: np.uint16}


class Db1DataError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise Db1DataError(f"{_LIB_PATH} is missing: run `python -m bdm_db1_amd.build` (there is no Python fallback)")
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, p = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER
        L.db1_idx_open.argtypes, L.db1_idx_open.restype = [ctypes.c_char_p, p(vp)], ctypes.c_int
        L.db1_idx_close.argtypes, L.db1_idx_close.restype = [vp], None
        for name in ("db1_idx_len", "db1_idx_doc_count"):
            getattr(L, name).argtypes, getattr(L, name).restype = [vp], i64
        for name in ("db1_idx_dtype_code", "db1_idx_elem_size"):
            getattr(L, name).argtypes, getattr(L, name).restype = [vp], ctypes.c_int
        for name in ("db1_idx_sizes", "db1_idx_pointers", "db1_idx_doc_idx"):
            getattr(L, name).argtypes, getattr(L, name).restype = [vp], vp
        L.db1_idx_get.argtypes, L.db1_idx_get.restype = [vp, i64, i64, i64, p(vp), p(i64)], ctypes.c_int
        L.db1_build_sample_idx.argtypes, L.db1_build_sample_idx.restype = [vp, vp, i32, i32, i64, vp, p(i64)], ctypes.c_int
        L.db1_build_rl_sample_idx.argtypes, L.db1_build_rl_sample_idx.restype = [vp, i64, i32, vp, p(i64)], ctypes.c_int
        L.db1_build_blending_indices.argtypes, L.db1_build_blending_indices.restype = [vp, vp, vp, i32, i64], ctypes.c_int
        L.db1_data_last_error.restype = ctypes.c_char_p
        L.db1_data_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise Db1DataError(f"libdb1_data: status {rc}: {_load().db1_data_last_error().decode()}")


def _view(addr, count, dtype):
    """zero-copy array over `count` elements at `addr` (possibly unaligned, like the reference's np.frombuffer views)"""
    if count == 0:
        return np.empty(0, dtype=dtype)
    buf = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(addr)
    a = np.frombuffer(buf, dtype=dtype, count=count)
    a.flags.writeable = False
    return a


class MMapIndexedDataset:
    def __init__(self, path, skip_warmup=True):
        L = _load()
        self._path = path
        h = ctypes.c_void_p()
        _check(L.db1_idx_open(os.fsencode(path), ctypes.byref(h)))
        self._h = h
        self._len = int(L.db1_idx_len(h))
        self._dtype = DTYPES[int(L.db1_idx_dtype_code(h))]
        self._sizes = _view(L.db1_idx_sizes(h), self._len, np.int32)
        self._pointers = _view(L.db1_idx_pointers(h), self._len, np.int64)
        self._doc_idx = _view(L.db1_idx_doc_idx(h), int(L.db1_idx_doc_count(h)), np.int64)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            self._sizes = self._pointers = self._doc_idx = None
            _lib.db1_idx_close(h)

    def __getstate__(self):
        return self._path

    def __setstate__(self, state):
        self.__init__(state)

    def __len__(self):
        return self._len

    @property
    def dtype(self):
        return self._dtype

    @property
    def sizes(self):
        return self._sizes

    @property
    def doc_idx(self):
        return self._doc_idx

    def get_doc_idx(self):
        return self._doc_idx

    def set_doc_idx(self, doc_idx_):
        self._doc_idx = doc_idx_

    @property
    def supports_prefetch(self):
        return False

    def get(self, idx, offset=0, length=None):
        data, n = ctypes.c_void_p(), ctypes.c_int64()
        _check(_load().db1_idx_get(self._h, int(idx), int(offset), -1 if length is None else int(length), ctypes.byref(data), ctypes.byref(n)))
        return _view(data.value, n.value, self._dtype)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.get(int(idx))
        if isinstance(idx, slice):
            start, stop, step = idx.indices(len(self))
            if step != 1:
                raise ValueError("Slices into indexed_dataset must be contiguous")
            sizes = self._sizes[idx]
            if len(sizes) == 0:
                return []
            first = self.get(start)  # items are stored back to back: one view over the whole slice, then split
            total = int(sizes.sum())
            flat = _view(first.ctypes.data, total, self._dtype) if total else np.empty(0, self._dtype)
            return np.split(flat, list(accumulate(int(s) for s in sizes))[:-1])
        raise TypeError(f"index must be int or slice, not {type(idx).__name__}")

    @staticmethod
    def exists(path):
        return os.path.exists(path + ".idx") and os.path.exists(path + ".bin")


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def build_sample_idx(sizes, doc_idx, seq_length, num_epochs, tokens_per_epoch):
    """int32 [num_samples + 1, 2]: (index into doc_idx, offset) of every sample boundary (helpers.cpp:117-203)"""
    sizes, doc_idx = _i32(sizes), _i32(doc_idx)
    n = ctypes.c_int64()
    L = _load()
    _check(L.db1_build_sample_idx(sizes.ctypes.data, doc_idx.ctypes.data, int(seq_length), int(num_epochs), int(tokens_per_epoch), None, ctypes.byref(n)))
    out = np.empty((n.value, 2), dtype=np.int32)
    _check(L.db1_build_sample_idx(sizes.ctypes.data, doc_idx.ctypes.data, int(seq_length), int(num_epochs), int(tokens_per_epoch), out.ctypes.data, ctypes.byref(n)))
    return out


def build_rl_sample_idx(path_lengths, transition_num):
    """int32 [sum(len - 1), 3]: (path, start, min(start + transition_num, len)) (helpers.cpp:82-115)"""
    pl = _i32(path_lengths)
    n = ctypes.c_int64()
    L = _load()
    _check(L.db1_build_rl_sample_idx(pl.ctypes.data, pl.shape[0], int(transition_num), None, ctypes.byref(n)))
    out = np.empty((n.value, 3), dtype=np.int32)
    _check(L.db1_build_rl_sample_idx(pl.ctypes.data, pl.shape[0], int(transition_num), out.ctypes.data, ctypes.byref(n)))
    return out


def build_blending_indices(dataset_index, dataset_sample_index, weights, num_datasets, size, verbose=False):
    """fills the caller's uint8 / int64 arrays in place, like helpers.build_blending_indices (helpers.cpp:20-80)"""
    assert dataset_index.dtype == np.uint8 and dataset_sample_index.dtype == np.int64 and dataset_index.flags.c_contiguous
    w = np.ascontiguousarray(weights, dtype=np.float64)
    _check(_load().db1_build_blending_indices(dataset_index.ctypes.data, dataset_sample_index.ctypes.data, w.ctypes.data, int(num_datasets), int(size)))
