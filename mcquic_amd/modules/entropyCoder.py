"""Entropy coder beside the tensor path (reference: mcquic/modules/entropyCoder.py:15-154).

State: the per-level code-frequency EMA (`_freqEMA.{l}` in the state_dict).  From it: normalised frequencies,
16-bit quantized CDFs (`mcq_pmf_to_quantized_cdf`), and per image x level rANS streams over the flattened
`code[m, h, w]` with the group index selecting the CDF (`mcq_rans_encode_with_indexes` / `_decode_`): the host-side
C++ coder in csrc/rans.cpp, bit-compatible with the reference's `mcquic.rans` extension.

The reference snapshot's `compress` / `decompress` start with `raise NotImplementedError` (entropyCoder.py:107,140)
and the dead body is self-inconsistent (`CodeSize.m` is built from an int at :126 and iterated as a list at :146).
What is implemented here is the evident intent of that body -- symbols = code.flatten(), indexes = group id,
cdfs = pmfToQuantizedCDF(freq_g, 16), cdfSizes = k + 2, offsets = 0 -- with `CodeSize.m` as the typed list
(`mcquic/utils/specification.py:89`).  The byte streams are pinned at the rANS level (bit-equal to the compiled
reference extension, tests/test_entropy_coder.py), not at the `Compressor.compress` level, which the reference
cannot execute.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .. import _lib
from ..ops import tensor_version
from ..utils.specification import CodeSize


def _vp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def pmfToQuantizedCDF(pmf, precision: int = 16) -> List[int]:
    """Same contract as `mcquic.rans.pmfToQuantizedCDF` (list of floats -> list of k + 1 ints)."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    cdf = np.empty(p.shape[0] + 1, dtype=np.uint32)
    rc = _lib.load().mcq_pmf_to_quantized_cdf(_vp(p), p.shape[0], precision, _vp(cdf))
    if rc != 0:
        raise ValueError("Invalid `pmf`: negative / non-finite element, all zero, or more symbols than 2**precision")
    return cdf.tolist()


class _Tables:
    """CDFs of one level flattened for the C ABI: m CDFs of k + 1 entries each."""

    def __init__(self, cdfs: List[List[int]], k: int):
        m = len(cdfs)
        self.cdfs = np.ascontiguousarray(np.asarray(cdfs, dtype=np.uint32).reshape(-1))
        self.starts = np.ascontiguousarray(np.arange(m, dtype=np.int32) * (k + 1))
        self.sizes = np.full(m, k + 2, dtype=np.int32)           # the reference's convention (entropyCoder.py:121)
        self.lens = np.full(m, k + 1, dtype=np.int32)            # entries actually present per CDF (range checks in C)
        self.offsets = np.zeros(m, dtype=np.int32)
        self.m = m


def ransEncodeWithIndexes(symbols: np.ndarray, indexes: np.ndarray, t: _Tables) -> bytes:
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    cap = 4 * symbols.size + 64
    out = np.empty(cap, dtype=np.uint8)
    n = _lib.load().mcq_rans_encode_with_indexes(_vp(symbols), _vp(indexes), symbols.size, _vp(t.cdfs), _vp(t.starts),
                                                 _vp(t.sizes), _vp(t.lens), _vp(t.offsets), t.m, _vp(out), cap)
    if n < 0:
        raise RuntimeError(f"rANS encode failed ({n}): a symbol outside its CDF's range, or a malformed table")
    return out[:n].tobytes()


def ransDecodeWithIndexes(binary: bytes, indexes: np.ndarray, t: _Tables) -> np.ndarray:
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    buf = np.frombuffer(binary, dtype=np.uint8)
    out = np.empty(indexes.size, dtype=np.int32)
    rc = _lib.load().mcq_rans_decode_with_indexes(_vp(buf), buf.size, _vp(indexes), indexes.size, _vp(t.cdfs), _vp(t.starts),
                                                  _vp(t.sizes), _vp(t.lens), _vp(t.offsets), t.m, _vp(out))
    if rc != 0:
        raise RuntimeError("Got a truncated or malformed rANS stream.")
    return out


def hostThreads() -> int:
    """Host threads for the batched coder: the cores this process may use (affinity mask; MCQUIC_AMD_RANS_THREADS overrides)."""
    env = os.environ.get("MCQUIC_AMD_RANS_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))


CODER_OVERLAP = os.environ.get("MCQUIC_AMD_CODER_OVERLAP", "1") != "0"     # A/B switch: 0 = copy / code all levels at once (rounds 1-5)
_POOL = None                     # ONE host thread for the level-wise coder jobs of this process (module state: models are deep-copied)


def _hostWorker():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mcq-rans")
    return _POOL


def _forget_worker_after_fork():
    """A forked child inherits the executor object but not its thread: the first job would wait forever.  Start over there."""
    global _POOL
    _POOL = None


if hasattr(os, "register_at_fork"):
    os.register_at_fork(after_in_child=_forget_worker_after_fork)


def _poolThreads(threads: int, symbols: int) -> int:
    """Threads for one batched call: never more than one per 16 K symbols.  A pool thread costs ~40 us to start and join, a
    symbol ~25 ns: ten 768x512 images (61 K symbols on level 0, 15 K and 4 K below) code fastest on 2-4 threads (0.30 ms per
    `compress` against 0.84 ms on 32, tools/probes/rans_threads.py on the GPU box's host) and the small levels on the caller's."""
    return max(1, min(threads or hostThreads(), symbols // 16384))


def ransEncodeBatchWithIndexes(symbols: np.ndarray, indexes: np.ndarray, t: _Tables, threads: int = 0) -> List[bytes]:
    """symbols [n_streams, n] (one stream per row, all coded with the same `indexes[n]`) -> one byte string per row.
    ONE C call; rows are spread over a pool of host threads (mcq_rans_encode_batch_with_indexes)."""
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    ns, n = symbols.shape
    stride = 4 * n + 64
    out = np.empty((ns, stride), dtype=np.uint8)
    sizes = np.zeros(ns, dtype=np.int64)
    rc = _lib.load().mcq_rans_encode_batch_with_indexes(_vp(symbols), ns, n, _vp(indexes), _vp(t.cdfs), _vp(t.starts), _vp(t.sizes),
                                                        _vp(t.lens), _vp(t.offsets), t.m, _vp(out), stride, _vp(sizes),
                                                        _poolThreads(threads, symbols.size))
    if rc != 0:
        raise RuntimeError(f"rANS encode failed ({rc}): a code index outside [0, k), or a malformed table")
    return [out[i, :sizes[i]].tobytes() for i in range(ns)]


def ransDecodeBatchWithIndexes(binaries: List[bytes], indexes: np.ndarray, t: _Tables, threads: int = 0,
                               out: Optional[np.ndarray] = None) -> np.ndarray:
    """One byte string per stream -> int32 [n_streams, n]; ONE C call over a pool of host threads.  `out`: a C-contiguous
    int32 [n_streams, n] array to decode into (e.g. the view of a pinned staging tensor)."""
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    offs = np.zeros(len(binaries) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in binaries], out=offs[1:])
    buf = np.frombuffer(b"".join(binaries), dtype=np.uint8)
    if out is None:
        out = np.empty((len(binaries), indexes.size), dtype=np.int32)
    elif out.dtype != np.int32 or out.shape != (len(binaries), indexes.size) or not out.flags.c_contiguous:
        raise ValueError("ransDecodeBatchWithIndexes: `out` must be a C-contiguous int32 [n_streams, n] array")
    rc = _lib.load().mcq_rans_decode_batch_with_indexes(_vp(buf), _vp(offs), len(binaries), _vp(indexes), indexes.size, _vp(t.cdfs),
                                                        _vp(t.starts), _vp(t.sizes), _vp(t.lens), _vp(t.offsets), t.m, _vp(out),
                                                        _poolThreads(threads, out.size))
    if rc != 0:
        raise RuntimeError("Got a truncated or malformed rANS stream.")
    return out


class EntropyCoder(nn.Module):
    def __init__(self, m: int, k: List[int], ema: float = 0.9):
        super().__init__()
        # initial value is uniform (entropyCoder.py:22)
        self._freqEMA = nn.ParameterList(nn.Parameter(torch.ones(m, ki) / ki, requires_grad=False) for ki in k)
        self._m, self._k, self._ema = m, k, ema
        self._cdfs = None
        self._normalizedFreq = None
        self._tables = None
        self._key = None

    @torch.no_grad()
    def forward(self, codes: List[torch.Tensor], counts: Optional[torch.Tensor] = None):
        """Frequency EMA update of a training step (entropyCoder.py:28-44).  The reference sums one-hot codes
        [n, m, h, w, k] over (n, h, w) and all-reduces each level; here the counts are histograms of the int64 codes
        and the levels share ONE all-reduce.  `counts`: this rank's histograms of all levels back to back (int64), already
        made where the codes were (the sampling kernels add into `countBuffer()`); without it they are counted from `codes`."""
        from ..parallel import all_reduce_, local_code_counts
        if counts is None:
            counts = local_code_counts(codes, self._k)
            sink = self.__dict__.get("_countBuf")
            if self.__dict__.get("_countSinkOn", False) and sink is not None and sink.shape == counts.shape and sink.device == counts.device:
                sink.copy_(counts)
                counts = sink
        self.__dict__["_countMs"] = [int(c.shape[1]) for c in codes]
        if self.__dict__.get("_countSinkOn", False):
            self.__dict__["_countSinkBuf"] = counts           # deferred: the caller reduces it over the ranks and calls applyCounts
            return
        self._emaUpdate(all_reduce_(counts))

    def _levelMs(self) -> List[int]:
        return list(self._m) if isinstance(self._m, (list, tuple)) else [self._m] * len(self._k)

    def countBuffer(self, device) -> torch.Tensor:
        """This coder's code-count buffer on `device` (int64, level l's [m_l, k_l] histogram at sum_{j<l} m_j k_j): allocated
        once, so a captured training step keeps its address; zeroed by the step's prologue launch, added into by the sampling
        kernels, read by the EMA update (here or, deferred, after the caller's all-reduce)."""
        buf = self.__dict__.get("_countBuf")
        n = sum(m * k for m, k in zip(self._levelMs(), self._k))
        if buf is None or buf.device != torch.device(device) or buf.numel() != n:
            buf = self.__dict__["_countBuf"] = torch.empty(n, dtype=torch.int64, device=device)
        return buf

    def _emaUpdate(self, counts):
        """`counts`: the global histograms, flat (level after level) or as a list of [m_l, k_l] tensors."""
        if torch.is_tensor(counts) and counts.is_cuda:
            from .. import ops
            ops.freq_ema_update_([f for f in self._freqEMA], counts, self._ema)      # all levels in one launch, in place
        else:
            if torch.is_tensor(counts):
                from ..parallel import split_code_counts
                counts = split_code_counts(counts, self._levelMs(), self._k)
            for lv, totalCount in enumerate(counts):
                totalCount = totalCount.to(self._freqEMA[lv].dtype)
                normalized = totalCount / totalCount.sum(-1, keepdim=True)
                ema = (1 - self._ema) * normalized + self._ema * self._freqEMA[lv]
                self._freqEMA[lv].copy_(ema)
        self.resetFreqAndCDF()

    # ---- deferred mode (parallel.GraphedTrainStep): forward leaves this rank's counts in a static buffer, the caller
    #      all-reduces them outside the captured step and hands the global counts to applyCounts ------------------------
    def deferCounts(self, on: bool):
        self.__dict__["_countSinkOn"] = bool(on)
        if not on:
            self.__dict__["_countSinkBuf"] = None

    def countSink(self) -> torch.Tensor:
        buf = self.__dict__.get("_countSinkBuf")
        if buf is None:
            raise RuntimeError("countSink: no training forward has run in deferred mode yet")
        return buf

    @torch.no_grad()
    def applyCounts(self, flat: torch.Tensor):
        self._emaUpdate(flat)

    # ---- frequency / CDF tables (entropyCoder.py:46-79) ------------------------------------------------------
    def resetFreqAndCDF(self):
        self._normalizedFreq = None
        self._cdfs = None
        self._tables = None

    def _stale(self) -> bool:
        key = tuple((tensor_version(f), f.data_ptr()) for f in self._freqEMA)
        if key != self._key:
            self._key = key
            return True
        return self._cdfs is None

    def updateFreqAndCDF(self):
        freq = [(f / f.sum(-1, keepdim=True)).detach().clone() for f in self._freqEMA]
        cdfs = [[pmfToQuantizedCDF(frAtM.tolist(), 16) for frAtM in fr.cpu()] for fr in freq]
        self._normalizedFreq = freq
        self._cdfs = cdfs
        self._tables = [_Tables(c, ki) for c, ki in zip(cdfs, self._k)]

    @property
    def CDFs(self) -> List[List[List[int]]]:
        if self._stale():
            self.updateFreqAndCDF()
        return self._cdfs

    @property
    def NormalizedFreq(self) -> List[torch.Tensor]:
        """List of [m, k_l] probabilities."""
        if self._stale():
            self.updateFreqAndCDF()
        return self._normalizedFreq

    def _checkShape(self, codes: List[torch.Tensor]):
        """Same checks and RuntimeErrors as entropyCoder.py:79-93."""
        info = ("Please give codes with correct shape, for example, [[1, 2, 24, 24], [1, 2, 12, 12], ...], which is a "
                "`level` length list. each code has shape [n, m, h, w]. ")
        if len(codes) < 1:
            raise RuntimeError("Length of codes is 0.")
        n, m = codes[0].shape[0], codes[0].shape[1]
        for code in codes:
            newN, newM = code.shape[0], code.shape[1]
            if n < 1:
                raise RuntimeError(info + "Now `n` = 0.")
            if m != newM:
                raise RuntimeError(info + "Now `m` is inconsisitent.")
            if n != newN:
                raise RuntimeError(info + "Now `n` is inconsisitent.")
        return n, m

    # ---- byte streams (entropyCoder.py:95-154) ----------------------------------------------------------------
    @torch.inference_mode()
    def compress(self, codes: List[torch.Tensor]) -> Tuple[List[List[bytes]], List[CodeSize]]:
        """codes: level-length list of [n, m, h, w] -> (binaries[n][level], CodeSize per image).
        Per level: one device-to-host copy (all levels' copies are enqueued before the first one is waited for) and ONE
        call into the host coder, which spreads the n images' streams over a thread pool."""
        n, m = self._checkShape(codes)
        self.CDFs  # refresh the tables if the EMA changed
        if len(codes) != len(self._tables):
            raise RuntimeError(f"expected {len(self._tables)} code levels, got {len(codes)}")
        hosts = []
        for code in codes:
            c32 = code.detach().to(torch.int32)
            if c32.is_cuda:
                pinned = torch.empty(c32.shape, dtype=torch.int32, pin_memory=True)
                pinned.copy_(c32, non_blocking=True)
                c32 = pinned
            hosts.append(c32)
        if codes[0].is_cuda:
            torch.cuda.current_stream(codes[0].device).synchronize()
        compressed: List[List[bytes]] = [[] for _ in range(n)]
        heights, widths = [], []
        for host, table, k in zip(hosts, self._tables, self._k):
            _, _, h, w = host.shape
            heights.append(h)
            widths.append(w)
            idx = np.repeat(np.arange(m, dtype=np.int32), h * w)            # group id per symbol, [m, h, w] order
            for i, b in enumerate(ransEncodeBatchWithIndexes(host.numpy().reshape(n, m * h * w), idx, table)):
                compressed[i].append(b)
        return compressed, [CodeSize([m] * len(codes), heights, widths, list(self._k)) for _ in range(n)]

    # ---- the same, overlapped with the kernels that make / use the codes (round 6) ---------------------------------------------
    # `compress` above copies every level, waits for the whole stream, then codes on the host; `decompress` decodes every level
    # before the first kernel of the decoder is enqueued.  But level 0's codes (3/4 of the symbols) exist long before levels 1 .. are
    # computed, and the decoder needs the LAST level (the smallest stream) first: a level-wise interface lets the quantizer hand each
    # level over as soon as its `vq_assign` is enqueued (copy in stream order, coded on a host thread while the GPU works on the
    # next level), and take decoded levels one by one, smallest first, while a host thread decodes the larger ones.  Same bytes,
    # same codes (tests/test_gpu_entropy_coder.py).  MCQUIC_AMD_CODER_OVERLAP=0: the all-at-once forms above.
    def _worker(self):
        return _hostWorker()

    def beginCompress(self, levels: int) -> "_CompressJob":
        self.CDFs                                              # refresh the tables if the EMA changed
        if levels != len(self._tables):
            raise RuntimeError(f"expected {len(self._tables)} code levels, got {levels}")
        return _CompressJob(self)

    def beginDecompress(self, binaries: List[List[bytes]], codeSizes: List[CodeSize]) -> "_DecompressJob":
        first = self._checkHeaders(binaries, codeSizes)
        return _DecompressJob(self, binaries, first)

    # what a header may ask the decoder to allocate: a side of 2^15 latent positions is a 2-megapixel-wide image 64 times over
    _MAX_SIDE = 1 << 15
    _MAX_AREA = 1 << 24          # latent positions per level and image (16 M = a 64 k x 64 k pixel image at stride 16)
    _MAX_SYMBOLS = 1 << 28       # n * m * h * w per level: what one call may stage (1 GiB of int32)

    def _checkHeaders(self, binaries: List[List[bytes]], codeSizes: List[CodeSize]) -> CodeSize:
        """Header fields are untrusted input (they come out of a `.mcq` file): m / k must be this model's, sizes positive and
        bounded, all images of a batch alike -- checked before anything is allocated.  Returns the batch's (common) CodeSize."""
        if len(binaries) < 1 or len(binaries) != len(codeSizes):
            raise RuntimeError("`binaries` and `codeSizes` must be non-empty and of equal length.")
        self.CDFs
        levels = len(self._tables)
        first = codeSizes[0]
        for binary, codeSize in zip(binaries, codeSizes):
            if len(binary) != levels or len(codeSize.heights) != levels or len(codeSize.widths) != levels or len(codeSize.m) != levels:
                raise RuntimeError(f"Every image must carry one stream per level ({levels} for this model).")
            if list(codeSize.m) != [self._m] * levels or list(codeSize.k) != list(self._k):
                raise RuntimeError(f"The header's code size (m = {list(codeSize.m)}, k = {list(codeSize.k)}) is not this model's "
                                   f"(m = {self._m}, k = {list(self._k)}).")
            if any(not (0 < int(v) <= self._MAX_SIDE) for v in list(codeSize.heights) + list(codeSize.widths)):
                raise RuntimeError("The header's code heights / widths are out of range.")
            if list(codeSize.heights) != list(first.heights) or list(codeSize.widths) != list(first.widths):
                raise RuntimeError("All images of one batch must share their code sizes.")
        for lv in range(levels):
            h, w = int(first.heights[lv]), int(first.widths[lv])
            # the sides are bounded above, but so must be what they multiply to: h = w = 2^15 would stage 4 GiB * m per array
            if h * w > self._MAX_AREA or len(binaries) * self._m * h * w > self._MAX_SYMBOLS:
                raise RuntimeError("The header's code sizes ask for more symbols than a call may decode.")
            # every level halves the one before it, rounding up (stride-2 convs with padding 1: ceil(h / 2))
            if lv > 0 and (h != (int(first.heights[lv - 1]) + 1) // 2 or w != (int(first.widths[lv - 1]) + 1) // 2):
                raise RuntimeError("The header's code sizes do not halve from level to level.")
        return first

    @torch.inference_mode()
    def decompress(self, binaries: List[List[bytes]], codeSizes: List[CodeSize]) -> List[torch.Tensor]:
        """binaries[n][level] -> level-length list of int64 [n, m, h, w] on the coder's device.  Header fields are
        untrusted input (they come out of a `.mcq` file): m / k must be this model's, sizes positive and bounded, all
        images of a batch alike -- checked before anything is allocated."""
        first = self._checkHeaders(binaries, codeSizes)
        levels = len(self._tables)
        device = self._freqEMA[0].device
        n = len(binaries)
        out = []
        for lv, table in enumerate(self._tables):
            h, w = int(first.heights[lv]), int(first.widths[lv])
            idx = np.repeat(np.arange(self._m, dtype=np.int32), h * w)
            # decoded straight into a pinned staging tensor and handed to the device without waiting for it: the host is
            # never blocked behind the GPU's queue (a pageable copy is), so the coder of the next call runs while the GPU still
            # decodes this one; int32 -> int64 happens on the device
            host = torch.empty((n, idx.size), dtype=torch.int32, pin_memory=device.type == "cuda")
            ransDecodeBatchWithIndexes([binary[lv] for binary in binaries], idx, table, out=host.numpy())
            out.append(host.to(device, non_blocking=True).to(torch.int64).reshape(n, self._m, h, w))
        return out



class _CompressJob:
    """One `compress` call taken level by level (EntropyCoder.beginCompress): submit(level, code) right after the level's codes are
    enqueued; finish() -> (binaries[n][level], CodeSize per image)."""

    def __init__(self, coder: "EntropyCoder"):
        self.coder = coder
        self.futures = {}
        self.shapes = {}

    def submit(self, lv: int, code: torch.Tensor) -> None:
        coder = self.coder
        n, m, h, w = code.shape
        if m != coder._m:
            raise RuntimeError("Please give codes with correct shape: `m` is inconsisitent.")
        self.shapes[lv] = (n, m, h, w)
        c32 = code.detach().to(torch.int32)
        done = None
        if c32.is_cuda:
            # the copy stays on the COMPUTE stream (123 KB for ten images' level 0: a few microseconds in stream order) and an event
            # behind it releases the host thread.  A copy stream of its own shares a hardware queue with other streams once a process
            # has made a few (the bench's main process: side streams of the blocks, prefetch streams) and then waits behind the
            # kernels it was meant to overlap: compress fell from 219 to 183 Mpps there (profiles/r06_speed_protocol_overlap.txt)
            pinned = torch.empty(c32.shape, dtype=torch.int32, pin_memory=True)
            pinned.copy_(c32, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(c32.device))
            c32 = pinned
        table = coder._tables[lv]

        def code_level(host=c32, done=done, table=table, n=n, m=m, h=h, w=w):
            if done is not None:
                done.synchronize()
            idx = np.repeat(np.arange(m, dtype=np.int32), h * w)            # group id per symbol, [m, h, w] order
            return ransEncodeBatchWithIndexes(host.numpy().reshape(n, m * h * w), idx, table)
        self.futures[lv] = coder._worker().submit(code_level)

    def finish(self):
        coder = self.coder
        levels = len(coder._tables)
        if sorted(self.futures) != list(range(levels)):
            raise RuntimeError(f"expected {levels} code levels, got {len(self.futures)}")
        n, m = self.shapes[0][0], self.shapes[0][1]
        if any(self.shapes[lv][0] != n for lv in range(levels)):
            raise RuntimeError("Please give codes with correct shape: `n` is inconsisitent.")
        compressed: List[List[bytes]] = [[] for _ in range(n)]
        for lv in range(levels):
            for i, b in enumerate(self.futures[lv].result()):
                compressed[i].append(b)
        heights = [self.shapes[lv][2] for lv in range(levels)]
        widths = [self.shapes[lv][3] for lv in range(levels)]
        return compressed, [CodeSize([m] * levels, heights, widths, list(coder._k)) for _ in range(n)]


class _DecompressJob:
    """One `decompress` call taken level by level (EntropyCoder.beginDecompress): every level is decoded on the coder's host thread,
    LAST level first (the order the decoder cascade consumes them in); level(lv) -> int64 [n, m, h, w] on the coder's device."""

    def __init__(self, coder: "EntropyCoder", binaries: List[List[bytes]], first: CodeSize):
        self.coder = coder
        self.device = coder._freqEMA[0].device
        self.n = len(binaries)
        levels = len(coder._tables)
        self.futures = {}
        for lv in reversed(range(levels)):
            h, w = int(first.heights[lv]), int(first.widths[lv])
            streams = [binary[lv] for binary in binaries]
            self.futures[lv] = coder._worker().submit(self._decode_level, lv, streams, h, w)

    def _decode_level(self, lv, streams, h, w):
        coder = self.coder
        idx = np.repeat(np.arange(coder._m, dtype=np.int32), h * w)
        host = torch.empty((self.n, idx.size), dtype=torch.int32, pin_memory=self.device.type == "cuda")
        ransDecodeBatchWithIndexes(streams, idx, coder._tables[lv], out=host.numpy())
        return host, h, w

    def level(self, lv: int) -> torch.Tensor:
        host, h, w = self.futures[lv].result()
        return host.to(self.device, non_blocking=True).to(torch.int64).reshape(self.n, self.coder._m, h, w)


class VariousMCoder(nn.Module):
    """Code-frequency statistics + byte streams for quantizers whose group count differs per level (reference:
    mcquic/modules/entropyCoder.py:295-425, the coder `Neon` / `ResidualBackwardQuantizer` use).  Like the reference's
    live implementation the byte streams are the RAW int64 codes (`codePerImage.cpu().numpy().tobytes()`, :372; rANS is not
    applied there), so files written by either side read back on the other; `_freqEMA.{l}` [m_l, k_l], EMA 0.998."""

    def __init__(self, m: List[int], k: List[int], ema: float = 0.998):
        super().__init__()
        self._freqEMA = nn.ParameterList(nn.Parameter(torch.ones(mi, ki) / ki, requires_grad=False) for mi, ki in zip(m, k))
        self._k, self._m, self._ema = k, m, ema
        self._cdfs = None
        self._normalizedFreq = None

    # EMA update (:306-323) from integer codes (level l: [n, m_l, h, w]); one fused all-reduce for all levels
    forward = EntropyCoder.forward
    _levelMs = EntropyCoder._levelMs
    countBuffer = EntropyCoder.countBuffer
    _emaUpdate = EntropyCoder._emaUpdate
    deferCounts = EntropyCoder.deferCounts
    countSink = EntropyCoder.countSink
    applyCounts = EntropyCoder.applyCounts

    def resetFreqAndCDF(self):
        self._normalizedFreq = None
        self._cdfs = None

    def updateFreqAndCDF(self):
        freq = [(f / f.sum(-1, keepdim=True)).detach().clone() for f in self._freqEMA]
        self._cdfs = [[pmfToQuantizedCDF(frAtM.tolist(), 16) for frAtM in fr.cpu()] for fr in freq]
        self._normalizedFreq = freq

    @property
    def CDFs(self) -> List[List[List[int]]]:
        if self._cdfs is None or self._normalizedFreq is None:
            self.updateFreqAndCDF()
        return self._cdfs

    @property
    def NormalizedFreq(self) -> List[torch.Tensor]:
        if self._cdfs is None or self._normalizedFreq is None:
            self.updateFreqAndCDF()
        return self._normalizedFreq

    def _checkShape(self, codes: List[torch.Tensor]) -> int:
        """:348-361 (`m` may differ between levels here)."""
        info = ("Please give codes with correct shape, for example, [[1, 2, 24, 24], [1, 2, 12, 12], ...], which is a "
                "`level` length list. each code has shape [n, m, h, w]. ")
        if len(codes) < 1:
            raise RuntimeError("Length of codes is 0.")
        n = codes[0].shape[0]
        for code in codes:
            if n < 1:
                raise RuntimeError(info + "Now `n` = 0.")
            if n != code.shape[0]:
                raise RuntimeError(info + "Now `n` is inconsisitent.")
        return n

    @torch.inference_mode()
    def compress(self, codes: List[torch.Tensor]) -> Tuple[List[List[bytes]], List[CodeSize]]:
        n = self._checkShape(codes)
        compressed: List[List[bytes]] = [[] for _ in range(n)]
        heights, widths = [], []
        for code in codes:
            heights.append(code.shape[2])
            widths.append(code.shape[3])
            host = code.detach().to("cpu", torch.int64).contiguous().numpy()      # one D2H copy per level
            for i in range(n):
                compressed[i].append(host[i].tobytes())
        return compressed, [CodeSize(list(self._m), heights, widths, list(self._k)) for _ in range(n)]

    @torch.inference_mode()
    def decompress(self, binaries: List[List[bytes]], codeSizes: List[CodeSize]) -> List[torch.Tensor]:
        if len(binaries) < 1 or len(binaries) != len(codeSizes):
            raise RuntimeError("`binaries` and `codeSizes` must be non-empty and of equal length.")
        levels = len(binaries[0])
        out = [[] for _ in range(levels)]
        for binary, codeSize in zip(binaries, codeSizes):
            if len(binary) != levels or len(codeSize.heights) != levels or len(codeSize.m) != levels:
                raise RuntimeError("Every image must carry one stream per level.")
            for lv, (b, mi, h, w) in enumerate(zip(binary, codeSize.m, codeSize.heights, codeSize.widths)):
                if mi < 1 or h < 1 or w < 1 or len(b) != 8 * mi * h * w:
                    raise RuntimeError("A stream's length does not match its header's code size.")
                out[lv].append(torch.from_numpy(np.frombuffer(b, dtype=np.int64).copy()).reshape(mi, h, w))
        device = self._freqEMA[0].device
        return [torch.stack(c, 0).to(device) for c in out]
