"""ctypes loader for oracle/libmcq_oracle.so (the plain-C restatement; TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "libmcq_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", HERE, "libmcq_oracle.so"], check=True)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def conv2d(x, w, bias, stride=1, wide=False):
    x, w = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    n, cin, h, wd = x.shape
    cout, _, ks, _ = w.shape
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    y = np.empty((n, cout, ho, wo), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = load().mcq_oracle_conv2d(_p(x), _p(w), None if b is None else _p(b), _p(y), n, cin, h, wd, cout, ks, stride, int(wide))
    assert rc == 0
    return y


def vq_assign(x, cb):
    x, cb = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(cb, np.float32)
    m, k, d = cb.shape
    n, _, h, w = x.shape
    codes = np.empty((n, m, h, w), np.int64)
    gap = np.empty((n, m, h, w), np.float64)
    rc = load().mcq_oracle_vq_assign(_p(x), _p(cb), _p(codes), _p(gap), n, m, d, h, w, k)
    assert rc == 0
    return codes, gap


def vq_gather(codes, cb):
    codes, cb = np.ascontiguousarray(codes, np.int64), np.ascontiguousarray(cb, np.float32)
    m, k, d = cb.shape
    n, _, h, w = codes.shape
    out = np.empty((n, m * d, h, w), np.float32)
    rc = load().mcq_oracle_vq_gather(_p(codes), _p(cb), _p(out), n, m, d, h, w, k)
    assert rc == 0
    return out
