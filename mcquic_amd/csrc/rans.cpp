// Host-side entropy coder of the `.mcq` payload: range-ANS over code indices (CPU, plain C++17, no GPU work).
//
// Restates, behind a plain-C batch ABI, what the reference reaches through its pybind11 module `mcquic.rans`
// (CompressAI's cpp_exts over ryg_rans' 64-bit rANS):
//   pmf -> 16-bit quantized CDF           third_party/CompressAI/cpp_exts/ops.cpp:42-111  (pmfToQuantizedCDF)
//   encodeWithIndexes + flush             cpp_exts/buffered_rans_encoder.cpp:104-196, ryg_rans/rans64.h:77-103
//   decodeWithIndexes                     cpp_exts/rans_decoder.cpp:104-167, ryg_rans/rans64.h:107-142
// Stream format (must stay bit-identical so existing `.mcq` files decode): 64-bit state, L = 2^31, 32-bit words
// emitted backwards, symbols encoded in reverse, 16-bit probability precision, 4-bit bypass digits for values at
// or beyond the CDF's sentinel slot.  The reference's Python caller passes `cdfSizes = k + 2` for CDFs of k + 1
// entries (mcquic/modules/entropyCoder.py:121), i.e. sentinel = k, which no code index ever reaches: the bypass
// branch is kept for format completeness.
//
// Unlike the reference (Python lists -> std::vector copies, one call per image and level) the entry points take
// flat int32 arrays.  Streams are independent (image x level): the *_batch_* entry points code all images of one level
// in ONE call, spread over a pool of host threads (one stream per task, tasks handed out through an atomic counter).
// Every table access is range-checked against `cdf_lens` (entries actually present per CDF): with the reference's
// `cdfSizes = k + 2` over k + 1 entries a symbol outside [0, k) would index one past the table (the reference reads
// out of bounds there, assert compiled out) -- here it is MCQ_EINVAL.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mcquic_hip.h"

namespace {

constexpr int kPrecision = 16;
constexpr uint32_t kBypassBits = 4;
constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;
constexpr uint64_t kRansL = 1ull << 31;

struct Sym { uint16_t start; uint16_t range; bool bypass; };

inline void enc_put(uint64_t& x, uint32_t*& ptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t x_max = ((kRansL >> scale_bits) << 32) * freq;
    if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
    x = ((x / freq) << scale_bits) + (x % freq) + start;
}

inline void enc_put_bits(uint64_t& x, uint32_t*& ptr, uint32_t val, uint32_t nbits) {
    const uint32_t freq = 1u << (16 - nbits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
    x = (x << nbits) | val;
}

inline uint32_t dec_get_bits(uint64_t& x, const uint32_t*& ptr, const uint32_t* end, uint32_t nbits, bool& ok) {
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < kRansL) {
        if (ptr >= end) { ok = false; return 0; }
        x = (x << 32) | *ptr++;
    }
    return val;
}

}  // namespace

extern "C" int mcq_pmf_to_quantized_cdf(const float* pmf, int32_t k, int32_t precision, uint32_t* cdf) {
    if (!pmf || !cdf || k <= 0 || precision <= 0 || precision > 16) return MCQ_EINVAL;
    for (int i = 0; i < k; ++i)
        if (pmf[i] < 0 || !std::isfinite(pmf[i])) return MCQ_EINVAL;       // the reference throws std::domain_error
    cdf[0] = 0;
    for (int i = 0; i < k; ++i) cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << precision));
    uint32_t total = 0;
    for (int i = 0; i <= k; ++i) total += cdf[i];
    if (total == 0) return MCQ_EINVAL;
    for (int i = 0; i <= k; ++i) cdf[i] = (uint32_t)(((uint64_t)(1ull << precision) * cdf[i]) / total);
    for (int i = 1; i <= k; ++i) cdf[i] += cdf[i - 1];
    cdf[k] = 1u << precision;
    for (int i = 0; i < k; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            // steal one count from the least frequent symbol that can spare it
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < k; ++j) {
                const uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal < 0) return MCQ_EINVAL;                             // more symbols than probability mass
            if (best_steal < i) { for (int j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
    return MCQ_OK;
}

namespace {

struct Tables {
    const uint32_t* cdfs; const int32_t* starts; const int32_t* sizes; const int32_t* lens; const int32_t* offsets; int32_t n_cdfs;
    bool ok() const {
        if (!cdfs || !starts || !sizes || !lens || !offsets || n_cdfs <= 0) return false;
        for (int32_t c = 0; c < n_cdfs; ++c)
            if (starts[c] < 0 || lens[c] < 2 || sizes[c] < 2) return false;
        return true;
    }
};

int64_t encode_stream(const int32_t* symbols, const int32_t* indexes, int64_t n, const Tables& t, uint8_t* out, int64_t capacity) {
    const uint32_t* cdfs = t.cdfs; const int32_t* cdf_starts = t.starts; const int32_t* cdf_sizes = t.sizes;
    const int32_t* offsets = t.offsets; const int32_t n_cdfs = t.n_cdfs;
    std::vector<Sym> syms;
    syms.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) return MCQ_EINVAL;
        const uint32_t* cdf = cdfs + cdf_starts[ci];
        const int32_t max_value = cdf_sizes[ci] - 2;
        if (max_value < 0) return MCQ_EINVAL;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        if (value + 1 >= t.lens[ci]) return MCQ_EINVAL;         // no sentinel slot in this table: the symbol is out of range
        syms.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
        if (value == max_value) {                 // bypass: digit count, then the raw value in 4-bit digits
            int32_t n_bypass = 0;
            while ((raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= (int32_t)kMaxBypass) { syms.push_back({(uint16_t)kMaxBypass, (uint16_t)(kMaxBypass + 1), true}); val -= kMaxBypass; }
            syms.push_back({(uint16_t)val, (uint16_t)(val + 1), true});
            for (int32_t jd = 0; jd < n_bypass; ++jd) {
                const uint32_t d = (raw >> (jd * kBypassBits)) & kMaxBypass;
                syms.push_back({(uint16_t)d, (uint16_t)(d + 1), true});
            }
        }
    }
    std::vector<uint32_t> buf(syms.size() + 2, 0xCCu);
    uint32_t* ptr = buf.data() + buf.size();
    uint64_t x = kRansL;
    for (size_t i = syms.size(); i-- > 0;) {
        const Sym& s = syms[i];
        if (!s.bypass) {
            if (s.range == 0) return MCQ_EINVAL;
            enc_put(x, ptr, s.start, s.range, kPrecision);
        } else {
            enc_put_bits(x, ptr, s.start, kBypassBits);
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    const int64_t nbytes = (int64_t)((buf.data() + buf.size()) - ptr) * 4;
    if (nbytes > capacity) return MCQ_ETOOLARGE;
    std::memcpy(out, ptr, (size_t)nbytes);
    return nbytes;
}

int decode_stream(const uint8_t* in, int64_t nbytes, const int32_t* indexes, int64_t n, const Tables& t, int32_t* out_symbols) {
    const uint32_t* cdfs = t.cdfs; const int32_t* cdf_starts = t.starts; const int32_t* cdf_sizes = t.sizes;
    const int32_t* offsets = t.offsets; const int32_t n_cdfs = t.n_cdfs;
    if (nbytes < 8 || (nbytes & 3)) return MCQ_EINVAL;
    std::vector<uint32_t> words((size_t)nbytes / 4);
    std::memcpy(words.data(), in, (size_t)nbytes);
    const uint32_t* ptr = words.data();
    const uint32_t* end = words.data() + words.size();
    uint64_t x = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
    ptr += 2;
    bool ok = true;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) return MCQ_EINVAL;
        const uint32_t* cdf = cdfs + cdf_starts[ci];
        const int32_t max_value = cdf_sizes[ci] - 2;
        if (max_value < 0 || max_value >= t.lens[ci]) return MCQ_EINVAL;       // the bisection reads entries 0 .. max_value
        const uint32_t cum = (uint32_t)(x & ((1u << kPrecision) - 1));
        // first entry strictly above cum (the reference scans linearly; CDFs are increasing, so bisect)
        int32_t lo = 0, hi = max_value + 1;      // entries 0 .. max_value + 1 hold cdf[0] = 0 .. 2^16 (or the sentinel)
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (cdf[mid] > cum) hi = mid; else lo = mid + 1; }
        const int32_t s = lo - 1;
        if (s < 0 || s > max_value || s + 1 >= t.lens[ci]) return MCQ_EINVAL;
        const uint32_t start = cdf[s], freq = cdf[s + 1] - cdf[s];
        x = (uint64_t)freq * (x >> kPrecision) + (x & ((1u << kPrecision) - 1)) - start;
        if (x < kRansL) {
            if (ptr >= end) return MCQ_EINVAL;
            x = (x << 32) | *ptr++;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(x, ptr, end, kBypassBits, ok);
            int32_t n_bypass = val;
            while (ok && val == (int32_t)kMaxBypass) { val = (int32_t)dec_get_bits(x, ptr, end, kBypassBits, ok); n_bypass += val; }
            int32_t raw = 0;
            for (int32_t jd = 0; ok && jd < n_bypass; ++jd) raw |= (int32_t)dec_get_bits(x, ptr, end, kBypassBits, ok) << (jd * kBypassBits);
            if (!ok) return MCQ_EINVAL;
            value = raw >> 1;
            if (raw & 1) value = -value - 1; else value += max_value;
        }
        out_symbols[i] = value + offsets[ci];
    }
    return MCQ_OK;
}

// run task(i) for i in [0, n_tasks) on up to n_threads host threads; returns the first non-zero status in task order
template <typename F>
int run_pool(int64_t n_tasks, int32_t n_threads, F&& task) {
    if (n_tasks <= 0) return MCQ_OK;
    // Nothing may leave this function as an exception: inside a worker thread or across the extern "C" boundary that is
    // std::terminate for the whole Python process.  bad_alloc (stream sizes come out of file headers) and system_error from
    // thread creation (ulimit / cgroup limits) become MCQ_EINVAL / a smaller pool.
    try {
        std::vector<int> status((size_t)n_tasks, MCQ_OK);
        int64_t workers = n_threads <= 0 ? (int64_t)std::thread::hardware_concurrency() : n_threads;
        if (workers < 1) workers = 1;
        if (workers > n_tasks) workers = n_tasks;
        std::atomic<int64_t> next{0};
        auto loop = [&]() noexcept {
            for (;;) {
                const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= n_tasks) return;
                try { status[(size_t)i] = task(i); } catch (...) { status[(size_t)i] = MCQ_EINVAL; }
            }
        };
        std::vector<std::thread> pool;
        if (workers > 1) {
            try {
                pool.reserve((size_t)workers - 1);
                for (int64_t w = 1; w < workers; ++w) pool.emplace_back(loop);
            } catch (...) { /* fewer threads than asked for: the calling thread and whoever started share the tasks */ }
        }
        loop();
        for (auto& th : pool) th.join();
        for (int st : status) if (st != MCQ_OK) return st;
        return MCQ_OK;
    } catch (...) {
        return MCQ_EINVAL;
    }
}

}  // namespace

extern "C" int64_t mcq_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n,
                                                const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                                const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, uint8_t* out,
                                                int64_t capacity) {
    const Tables t{cdfs, cdf_starts, cdf_sizes, cdf_lens, offsets, n_cdfs};
    if (!symbols || !indexes || !out || n < 0 || !t.ok()) return MCQ_EINVAL;
    try { return encode_stream(symbols, indexes, n, t, out, capacity); } catch (...) { return MCQ_EINVAL; }
}

extern "C" int mcq_rans_decode_with_indexes(const uint8_t* in, int64_t nbytes, const int32_t* indexes, int64_t n,
                                            const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                            const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, int32_t* out_symbols) {
    const Tables t{cdfs, cdf_starts, cdf_sizes, cdf_lens, offsets, n_cdfs};
    if (!in || !indexes || !out_symbols || n < 0 || !t.ok()) return MCQ_EINVAL;
    try { return decode_stream(in, nbytes, indexes, n, t, out_symbols); } catch (...) { return MCQ_EINVAL; }
}

extern "C" int mcq_rans_encode_batch_with_indexes(const int32_t* symbols, int64_t n_streams, int64_t n, const int32_t* indexes,
                                                  const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                                  const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, uint8_t* out,
                                                  int64_t stride, int64_t* out_nbytes, int32_t n_threads) {
    const Tables t{cdfs, cdf_starts, cdf_sizes, cdf_lens, offsets, n_cdfs};
    if (!symbols || !indexes || !out || !out_nbytes || n_streams < 0 || n < 0 || stride < 8 || !t.ok()) return MCQ_EINVAL;
    return run_pool(n_streams, n_threads, [&](int64_t i) -> int {
        const int64_t nb = encode_stream(symbols + i * n, indexes, n, t, out + i * stride, stride);
        out_nbytes[i] = nb < 0 ? 0 : nb;
        return nb < 0 ? (int)nb : MCQ_OK;
    });
}

extern "C" int mcq_rans_decode_batch_with_indexes(const uint8_t* in, const int64_t* in_offsets, int64_t n_streams,
                                                  const int32_t* indexes, int64_t n, const uint32_t* cdfs, const int32_t* cdf_starts,
                                                  const int32_t* cdf_sizes, const int32_t* cdf_lens, const int32_t* offsets,
                                                  int32_t n_cdfs, int32_t* out_symbols, int32_t n_threads) {
    const Tables t{cdfs, cdf_starts, cdf_sizes, cdf_lens, offsets, n_cdfs};
    if (!in || !in_offsets || !indexes || !out_symbols || n_streams < 0 || n < 0 || !t.ok()) return MCQ_EINVAL;
    for (int64_t i = 0; i < n_streams; ++i)
        if (in_offsets[i + 1] < in_offsets[i]) return MCQ_EINVAL;
    return run_pool(n_streams, n_threads, [&](int64_t i) -> int {
        return decode_stream(in + in_offsets[i], in_offsets[i + 1] - in_offsets[i], indexes, n, t, out_symbols + i * n);
    });
}
