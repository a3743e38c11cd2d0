// Synthetic haystack generators (include/daac_synth.h): pure functions of (seed, byte index),
// evaluated on the device so multi-GiB haystacks never cross PCIe.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/daac_synth.h"
#include "pma.hpp"

namespace daac {

__host__ __device__ static inline uint64_t synth_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t synth_z(uint64_t seed, uint64_t j) {
    return synth_mix64(seed + (j + 1) * 0x9e3779b97f4a7c15ull);
}

struct Alphabet { uint8_t sym[256]; uint32_t n; };

__global__ __launch_bounds__(256) void synth_uniform_kernel(uint8_t *out, uint64_t len, uint64_t seed, Alphabet al, uint64_t off) {
    // one thread per 8-byte group of the GLOBAL stream, so shards agree with the whole
    const uint64_t g0 = off >> 3;
    const uint64_t ngroups = ((off + len + 7) >> 3) - g0;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ngroups;
         t += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t g = g0 + t;
        const uint64_t z = synth_z(seed, g);
        uint64_t packed = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t r = static_cast<uint32_t>(z >> (8 * b)) & 0xffu;
            packed |= static_cast<uint64_t>(al.sym[(r * al.n) >> 8]) << (8 * b);
        }
        const uint64_t first = g << 3;  // global index of byte 0 of this group
        if (first >= off && first + 8 <= off + len && ((reinterpret_cast<uintptr_t>(out) + (first - off)) & 7u) == 0) {
            *reinterpret_cast<uint64_t *>(out + (first - off)) = packed;
        } else {
            for (int b = 0; b < 8; ++b) {
                const uint64_t i = first + b;
                if (i >= off && i < off + len) out[i - off] = static_cast<uint8_t>(packed >> (8 * b));
            }
        }
    }
}

__global__ __launch_bounds__(256) void synth_wordsoup_kernel(uint8_t *out, uint64_t len, uint64_t seed, const uint8_t *words,
                                                               const uint64_t *offsets, uint32_t n_words, uint32_t slot, uint8_t pad,
                                                               uint32_t noise, Alphabet al, uint64_t off) {
    const uint64_t s0 = off / slot;
    const uint64_t nslots = (off + len + slot - 1) / slot - s0;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < nslots;
         t += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t s = s0 + t;
        const uint64_t z = synth_z(seed, s);
        const uint64_t first = s * slot;
        const bool is_noise = (z & 0xffu) < noise;
        const uint64_t w = (z >> 8) % n_words;
        const uint64_t wb = is_noise ? 0 : offsets[w];
        const uint32_t wl = is_noise ? slot - 1 : static_cast<uint32_t>(offsets[w + 1] - wb);
        uint64_t zz = 0;
        for (uint32_t j = 0; j < slot; ++j) {
            uint8_t c = pad;
            if (j < wl) {
                if (is_noise) {
                    if ((j & 7u) == 0) zz = synth_mix64(z + (j >> 3) + 1);
                    const uint32_t r = static_cast<uint32_t>(zz >> (8 * (j & 7u))) & 0xffu;
                    c = al.sym[(r * al.n) >> 8];
                } else {
                    c = words[wb + j];
                }
            }
            const uint64_t i = first + j;
            if (i >= off && i < off + len) out[i - off] = c;
        }
    }
}

// Zipf text (cfg5): the haystack is a sequence of `slot`-byte slots of whole UTF-8 characters.  Character k
// of slot s is drawn from zz = mix64(z(s) + k + 1): an ASCII byte ascii_lo + ((zz >> 8 & 0xffffff) * ascii_n >> 24) when
// (zz & 0xff) < ascii_256, otherwise the 3-byte code point cps[i], i = first index with cum[i] > ((zz >> 32) * cum[n-1]) >> 32.
// A slot is filled while 3 bytes are left; the last 1-2 bytes are ASCII draws of the same stream.
__global__ __launch_bounds__(256) void synth_zipf_text_kernel(uint8_t *out, uint64_t len, uint64_t seed, const uint32_t *cps,
                                                                const uint32_t *cum, uint32_t n_sym, uint32_t ascii_256,
                                                                uint32_t ascii_lo, uint32_t ascii_n, uint32_t slot, uint64_t off) {
    const uint64_t s0 = off / slot;
    const uint64_t nslots = (off + len + slot - 1) / slot - s0;
    const uint64_t total = cum[n_sym - 1];
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < nslots;
         t += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t s = s0 + t;
        const uint64_t z = synth_z(seed, s);
        const uint64_t first = s * slot;
        uint32_t j = 0;
        for (uint32_t k = 0; j < slot; ++k) {
            const uint64_t zz = synth_mix64(z + k + 1);
            uint8_t b[3];
            uint32_t nb = 1;
            if (slot - j < 3 || (zz & 0xffu) < ascii_256) {
                b[0] = static_cast<uint8_t>(ascii_lo + ((((zz >> 8) & 0xffffffu) * ascii_n) >> 24));
            } else {
                const uint32_t u = static_cast<uint32_t>(((zz >> 32) * total) >> 32);
                uint32_t lo = 0, hi = n_sym - 1;  // first index with cum[i] > u
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (cum[mid] > u) hi = mid; else lo = mid + 1;
                }
                const uint32_t cp = cps[lo];  // U+0800 .. U+FFFF
                b[0] = static_cast<uint8_t>(0xe0u | (cp >> 12));
                b[1] = static_cast<uint8_t>(0x80u | ((cp >> 6) & 0x3fu));
                b[2] = static_cast<uint8_t>(0x80u | (cp & 0x3fu));
                nb = 3;
            }
            for (uint32_t q = 0; q < nb; ++q, ++j) {
                const uint64_t i = first + j;
                if (i >= off && i < off + len) out[i - off] = b[q];
            }
        }
    }
}

}  // namespace daac

using namespace daac;

static daac_status synth_fail(hipError_t e, const char *what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return DAAC_ERR_DEVICE;
}

static bool make_alphabet(const uint8_t *alphabet, uint32_t n, Alphabet &al) {
    if (!alphabet || n == 0 || n > 256) { set_error("alphabet must have 1..256 symbols"); return false; }
    for (uint32_t i = 0; i < 256; ++i) al.sym[i] = alphabet[i < n ? i : 0];
    al.n = n;
    return true;
}

extern "C" daac_status daac_synth_uniform(uint8_t *dev_out, size_t len, uint64_t seed, const uint8_t *alphabet, uint32_t alphabet_len,
                                          uint64_t index_offset, void *stream) {
    Alphabet al;
    if (!make_alphabet(alphabet, alphabet_len, al)) return DAAC_ERR_INVALID_ARGUMENT;
    if (len == 0) return DAAC_OK;
    hipLaunchKernelGGL(synth_uniform_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), dev_out,
                       static_cast<uint64_t>(len), seed, al, index_offset);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DAAC_OK : synth_fail(e, "synth_uniform_kernel");
}

extern "C" daac_status daac_synth_wordsoup(uint8_t *dev_out, size_t len, uint64_t seed, const uint8_t *words, const uint64_t *offsets,
                                           uint32_t n_words, uint32_t slot_bytes, uint8_t pad, uint32_t noise_256,
                                           const uint8_t *alphabet, uint32_t alphabet_len, uint64_t index_offset, void *stream) {
    Alphabet al;
    if (!make_alphabet(alphabet, alphabet_len, al)) return DAAC_ERR_INVALID_ARGUMENT;
    if (!words || !offsets || n_words == 0 || slot_bytes < 2) { set_error("bad word list"); return DAAC_ERR_INVALID_ARGUMENT; }
    for (uint32_t i = 0; i < n_words; ++i)
        if (offsets[i + 1] - offsets[i] >= slot_bytes) { set_error("word does not fit its slot"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (len == 0) return DAAC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t *d_words = nullptr;
    uint64_t *d_offs = nullptr;
    const size_t wbytes = offsets[n_words];
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_words), wbytes + 16)) != hipSuccess) return synth_fail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_offs), (n_words + 1) * sizeof(uint64_t))) != hipSuccess) { (void)hipFree(d_words); return synth_fail(e, "hipMalloc"); }
    e = hipMemcpyAsync(d_words, words, wbytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_offs, offsets, (n_words + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(synth_wordsoup_kernel, dim3(4096), dim3(256), 0, s, dev_out, static_cast<uint64_t>(len), seed, d_words, d_offs,
                           n_words, slot_bytes, pad, noise_256, al, index_offset);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_words);
    (void)hipFree(d_offs);
    return e == hipSuccess ? DAAC_OK : synth_fail(e, "synth_wordsoup");
}

extern "C" daac_status daac_synth_zipf_text(uint8_t *dev_out, size_t len, uint64_t seed, const uint32_t *codepoints, const uint32_t *cum_weights,
                                            uint32_t n_symbols, uint32_t ascii_256, uint32_t ascii_lo, uint32_t ascii_n, uint32_t slot_bytes,
                                            uint64_t index_offset, void *stream) {
    if (!codepoints || !cum_weights || n_symbols == 0 || slot_bytes < 3 || ascii_n == 0 || ascii_lo + ascii_n > 128 || ascii_256 > 256) {
        set_error("bad zipf text parameters");
        return DAAC_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t i = 0; i < n_symbols; ++i)
        if (codepoints[i] < 0x800u || codepoints[i] > 0xffffu || (codepoints[i] >= 0xd800u && codepoints[i] < 0xe000u) ||
            (i && cum_weights[i] <= cum_weights[i - 1])) { set_error("code points must be 3-byte scalars with increasing cumulative weights"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (len == 0) return DAAC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *d = nullptr;
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void **>(&d), 2 * static_cast<size_t>(n_symbols) * sizeof(uint32_t))) != hipSuccess) return synth_fail(e, "hipMalloc");
    e = hipMemcpyAsync(d, codepoints, n_symbols * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n_symbols, cum_weights, n_symbols * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(synth_zipf_text_kernel, dim3(4096), dim3(256), 0, s, dev_out, static_cast<uint64_t>(len), seed, d, d + n_symbols,
                           n_symbols, ascii_256, ascii_lo, ascii_n, slot_bytes, index_offset);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    return e == hipSuccess ? DAAC_OK : synth_fail(e, "synth_zipf_text");
}
