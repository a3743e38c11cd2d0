// Microbenchmarks behind DESIGN.md's budget for the GRAM step (gfx950): issue rate of the VALU instructions the step is
// made of, LDS lookup cost by access shape (conflict-free byte tables, random 4/6/8/16-byte entries, aligned and
// unaligned), LDS store cost by active lanes, cross-lane moves.  All loops are inline asm so that the measured
// instruction is the one named.  16 waves per CU as in the scan kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pipes_bench tools/micro/pipes_bench.hip && /tmp/pipes_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static int g_cus = 256;
static double g_clk = 2.4e9;

// ------------------------------------------------------------------------------------------- VALU issue rates
// 8 independent chains per lane; OPS instructions per inner iteration
#define VALU_KERNEL(NAME, ASM, CONSTRAINT_EXTRA)                                                              \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint32_t seed, int iters) {                  \
        uint32_t a0 = threadIdx.x * 977u + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t s = seed | 1u;                                                                               \
        for (int i = 0; i < iters; ++i) {                                                                     \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(s) CONSTRAINT_EXTRA);                                                          \
        }                                                                                                     \
        uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                   \
        if (r == 0x12345u) out[0] = r;                                                                        \
    }

#define A_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define A_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define A_BFEV(k) "v_bfe_u32 %" #k ", %" #k ", %8, 1\n"
#define A_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 5, %8\n"
#define A_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %8\n"
#define A_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %8\n"
#define A_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 7\n"
#define A_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %8\n"
#define A_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %8\n"
#define A_FFBL(k) "v_ffbl_b32 %" #k ", %" #k "\n"
#define A_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define A_MOVSDWA(k) "v_mov_b32_sdwa %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define A_ADDSDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
#define A_LSHLSDWA(k) "v_lshlrev_b32_sdwa %" #k ", 2, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define A_MOVDPP(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_MOVWSHR(k) "v_mov_b32_dpp %" #k ", %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_ADDDPP(k) "v_add_u32_dpp %" #k ", %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define A_FADD(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define A_FMUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define A_FMAC(k) "v_fmac_f32 %" #k ", %8, %8\n"
#define A_CVTUB0(k) "v_cvt_f32_ubyte0 %" #k ", %" #k "\n"
#define A_CVTUB2(k) "v_cvt_f32_ubyte2 %" #k ", %" #k "\n"
#define A_CVTU32(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
#define A_CVTF32(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define A_PKMAD16(k) "v_pk_mad_u16 %" #k ", %" #k ", %8, %8\n"
#define A_PKADD16(k) "v_pk_add_u16 %" #k ", %" #k ", %8\n"
#define A_PKLSHL16(k) "v_pk_lshlrev_b16 %" #k ", 2, %" #k "\n"
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define A_CNDMASKS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define A_CMPCND(k) "v_cmp_ne_u32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define A_CMPSCND(k) "v_cmp_ne_u32 s[20:21], %" #k ", %8\nv_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define A_CMP(k) "v_cmp_ne_u32 vcc, %" #k ", %8\n"
#define A_CMPS(k) "v_cmp_ne_u32 s[20:21], %" #k ", %8\n"
#define A_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %8, %" #k "\n"
#define A_DOT4(k) "v_dot4_u32_u8 %" #k ", %" #k ", %8, %8\n"
#define A_SAD(k) "v_sad_u8 %" #k ", %" #k ", %8, %8\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define A_MAXU(k) "v_max_u32 %" #k ", %" #k ", %8\n"
#define A_XAD(k) "v_xad_u32 %" #k ", %" #k ", %8, %8\n"
#define A_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %8\n"
#define A_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %8\n"
#define A_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %8, %8\n"
#define A_READLANE(k) "v_readlane_b32 s20, %" #k ", 5\n"
#define A_MADU16(k) "v_mad_u16 %" #k ", %" #k ", %8, %8\n"
// round 5: what a packed-u16 main path of the count kernel would be made of
#define A_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define A_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define A_MINU(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define A_PKSUB16(k) "v_pk_sub_u16 %" #k ", %" #k ", %8\n"
#define A_PKMIN16(k) "v_pk_min_u16 %" #k ", %" #k ", %8\n"
#define A_MADU32U16(k) "v_mad_u32_u16 %" #k ", %" #k ", %8, %8 op_sel:[1,0,0,0]\n"
#define A_DOT2U16(k) "v_dot2_u32_u16 %" #k ", %" #k ", %8, %8\n"
#define A_LSHRSDWA(k) "v_lshrrev_b32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define A_LSHRV(k) "v_lshrrev_b32 %" #k ", %8, %" #k "\n"
#define A_SUBSDWA(k) "v_sub_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
#define A_ANDK(k) "v_and_b32 %" #k ", 0xff00ff, %" #k "\n"
#define NOCLOB
#define CLOB_VCC : "vcc"
#define CLOB_S : "s20", "s21"

VALU_KERNEL(k_add, A_ADD, NOCLOB)
VALU_KERNEL(k_and, A_AND, NOCLOB)
VALU_KERNEL(k_lshl, A_LSHL, NOCLOB)
VALU_KERNEL(k_bfe, A_BFE, NOCLOB)
VALU_KERNEL(k_bfev, A_BFEV, NOCLOB)
VALU_KERNEL(k_lshlor, A_LSHLOR, NOCLOB)
VALU_KERNEL(k_mad24, A_MAD24, NOCLOB)
VALU_KERNEL(k_add3, A_ADD3, NOCLOB)
VALU_KERNEL(k_alignbit, A_ALIGNBIT, NOCLOB)
VALU_KERNEL(k_perm, A_PERM, NOCLOB)
VALU_KERNEL(k_bcnt, A_BCNT, NOCLOB)
VALU_KERNEL(k_ffbl, A_FFBL, NOCLOB)
VALU_KERNEL(k_mov, A_MOV, NOCLOB)
VALU_KERNEL(k_movsdwa, A_MOVSDWA, NOCLOB)
VALU_KERNEL(k_addsdwa, A_ADDSDWA, NOCLOB)
VALU_KERNEL(k_lshlsdwa, A_LSHLSDWA, NOCLOB)
VALU_KERNEL(k_movdpp, A_MOVDPP, NOCLOB)
VALU_KERNEL(k_movwshr, A_MOVWSHR, NOCLOB)
VALU_KERNEL(k_adddpp, A_ADDDPP, NOCLOB)
VALU_KERNEL(k_fma, A_FMA, NOCLOB)
VALU_KERNEL(k_fadd, A_FADD, NOCLOB)
VALU_KERNEL(k_fmul, A_FMUL, NOCLOB)
VALU_KERNEL(k_fmac, A_FMAC, NOCLOB)
VALU_KERNEL(k_cvtub0, A_CVTUB0, NOCLOB)
VALU_KERNEL(k_cvtub2, A_CVTUB2, NOCLOB)
VALU_KERNEL(k_cvtu32, A_CVTU32, NOCLOB)
VALU_KERNEL(k_cvtf32, A_CVTF32, NOCLOB)
VALU_KERNEL(k_pkmad16, A_PKMAD16, NOCLOB)
VALU_KERNEL(k_pkadd16, A_PKADD16, NOCLOB)
VALU_KERNEL(k_pklshl16, A_PKLSHL16, NOCLOB)
VALU_KERNEL(k_lshladd, A_LSHLADD, NOCLOB)
VALU_KERNEL(k_cndmask, A_CNDMASK, CLOB_VCC)
VALU_KERNEL(k_cndmask_s, A_CNDMASKS, CLOB_S)
VALU_KERNEL(k_cmp_cnd, A_CMPCND, CLOB_VCC)
VALU_KERNEL(k_cmps_cnd, A_CMPSCND, CLOB_S)
VALU_KERNEL(k_cmp, A_CMP, CLOB_VCC)
VALU_KERNEL(k_cmps, A_CMPS, CLOB_S)
VALU_KERNEL(k_mbcnt, A_MBCNT, NOCLOB)
VALU_KERNEL(k_dot4, A_DOT4, NOCLOB)
VALU_KERNEL(k_sad, A_SAD, NOCLOB)
VALU_KERNEL(k_mullo, A_MULLO, NOCLOB)
VALU_KERNEL(k_maxu, A_MAXU, NOCLOB)
VALU_KERNEL(k_xad, A_XAD, NOCLOB)
VALU_KERNEL(k_andor, A_ANDOR, NOCLOB)
VALU_KERNEL(k_or3, A_OR3, NOCLOB)
VALU_KERNEL(k_bfi, A_BFI, NOCLOB)
VALU_KERNEL(k_readlane, A_READLANE, CLOB_S)
VALU_KERNEL(k_madu16, A_MADU16, NOCLOB)
VALU_KERNEL(k_sub, A_SUB, NOCLOB)
VALU_KERNEL(k_or, A_OR, NOCLOB)
VALU_KERNEL(k_xor, A_XOR, NOCLOB)
VALU_KERNEL(k_minu, A_MINU, NOCLOB)
VALU_KERNEL(k_pksub16, A_PKSUB16, NOCLOB)
VALU_KERNEL(k_pkmin16, A_PKMIN16, NOCLOB)
VALU_KERNEL(k_madu32u16, A_MADU32U16, NOCLOB)
VALU_KERNEL(k_dot2u16, A_DOT2U16, NOCLOB)
VALU_KERNEL(k_lshrsdwa, A_LSHRSDWA, NOCLOB)
VALU_KERNEL(k_lshrv, A_LSHRV, NOCLOB)
VALU_KERNEL(k_subsdwa, A_SUBSDWA, NOCLOB)
VALU_KERNEL(k_andk, A_ANDK, NOCLOB)

// packed f32 (register pairs) and 64-bit adds
__global__ __launch_bounds__(1024) void k_pkfma(uint32_t *out, uint32_t seed, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 s = {1.0001f, 0.5f};
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\n"
                     "v_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    }
    f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r.x + r.y == 12345.f) out[0] = 1;
}
__global__ __launch_bounds__(1024) void k_lshladd64(uint32_t *out, uint32_t seed, int iters) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint64_t s = seed * 0x100000001ull;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\n"
                     "v_lshl_add_u64 %4, %4, 0, %8\nv_lshl_add_u64 %5, %5, 0, %8\nv_lshl_add_u64 %6, %6, 0, %8\nv_lshl_add_u64 %7, %7, 0, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    }
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345u) out[0] = 1;
}

// v_mad_u64_u32: what the compiler picks for `a * b + c` on 32-bit values it cannot prove to be 24-bit (low half = the 32-bit result)
__global__ __launch_bounds__(1024) void k_mad64(uint32_t *out, uint32_t seed, int iters) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t s = seed | 1u;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %8, %0\nv_mad_u64_u32 %1, s[20:21], %8, %8, %1\nv_mad_u64_u32 %2, s[20:21], %8, %8, %2\n"
                     "v_mad_u64_u32 %3, s[20:21], %8, %8, %3\nv_mad_u64_u32 %4, s[20:21], %8, %8, %4\nv_mad_u64_u32 %5, s[20:21], %8, %8, %5\n"
                     "v_mad_u64_u32 %6, s[20:21], %8, %8, %6\nv_mad_u64_u32 %7, s[20:21], %8, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s20", "s21");
    }
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345u) out[0] = 1;
}

typedef void (*valu_fn)(uint32_t *, uint32_t, int);
static int run_valu(const char *name, valu_fn fn, int ops_per_asm = 8) {
    uint32_t *out;
    CHECK(hipMalloc(&out, 4));
    const int iters = 2048, blocks = g_cus * 2, threads = 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, 3u, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, 3u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = double(blocks) * threads / 64 * iters * ops_per_asm;
    printf("VALU %-26s %8.1f G wave-instr/s   %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, winst / ms / 1e6, g_cus * 4 * g_clk / (winst / ms * 1e3));
    CHECK(hipFree(out));
    return 0;
}

// ------------------------------------------------------------------------------------------- LDS lookups
// MODE: 0 ds_read_u8 (table of `span` bytes), 1 ds_read_u16, 2 ds_read_b32, 3 ds_read_b64 aligned, 4 ds_read_b128,
//       5 ds_read_b64 at 6-byte stride (unaligned), 6 ds_read_b64 at 5-byte stride, 7 ds_read2_b32 (two dwords 4 apart),
//       8 ds_read_b96 ... (not used)
// Addresses come from a per-lane LCG held in registers (cheap VALU, not the bottleneck: 3 VALU per lookup).
template <int MODE>
__global__ __launch_bounds__(1024) void lds_read_kernel(uint32_t *out, uint32_t span_entries, uint32_t stride, int iters, uint32_t nsym) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx = nsym ? (((x >> 16) * nsym) >> 16) : __umulhi(x, span_entries);
            uint32_t addr = idx * stride;
            uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (MODE == 0) asm volatile("ds_read_u8 %0, %1" : "=v"(v0) : "v"(addr));
            if (MODE == 1) asm volatile("ds_read_u16 %0, %1" : "=v"(v0) : "v"(addr));
            if (MODE == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(addr));
            if (MODE == 3 || MODE == 5 || MODE == 6) { uint64_t q; asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(addr)); v0 = uint32_t(q); v1 = uint32_t(q >> 32); }
            if (MODE == 4) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr)); v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w; }
            if (MODE == 7) { uint64_t q; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(q) : "v"(addr)); v0 = uint32_t(q); v1 = uint32_t(q >> 32); }
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            acc += v0 ^ v1 ^ v2 ^ v3;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345u) out[0] = acc;
}

// the same loop without the LDS instruction: the cost of the address generation alone
__global__ __launch_bounds__(1024) void lds_addr_only_kernel(uint32_t *out, uint32_t span_entries, uint32_t stride, int iters, uint32_t nsym) {
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx = nsym ? (((x >> 16) * nsym) >> 16) : __umulhi(x, span_entries);
            acc += idx * stride;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <class K>
static int time_kernel(K launch, double *ms_out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms;
    return 0;
}

template <int MODE>
static int run_lds(const char *name, uint32_t span_entries, uint32_t stride, uint32_t nsym, uint32_t *out) {
    const int iters = 512, blocks = g_cus, threads = 1024;
    const size_t lds = 160 * 1024 - 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_read_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    double ms = 0, ms0 = 0;
    if (time_kernel([&] { hipLaunchKernelGGL(lds_read_kernel<MODE>, dim3(blocks), dim3(threads), lds, 0, out, span_entries, stride, iters, nsym); }, &ms)) return 1;
    if (time_kernel([&] { hipLaunchKernelGGL(lds_addr_only_kernel, dim3(blocks), dim3(threads), 0, 0, out, span_entries, stride, iters, nsym); }, &ms0)) return 1;
    const double winst = double(blocks) * threads / 64 * iters * 8;
    const double cyc = (ms * 1e-3) * g_clk / (winst / g_cus);          // CU cycles per wave-instruction (all 16 waves of the CU share one LDS)
    const double cyc0 = (ms0 * 1e-3) * g_clk / (winst / g_cus);
    printf("LDS  %-44s %6.2f CU-cycles per wave-lookup (address math alone %5.2f)  -> %6.1f G lookups/s/chip\n", name, cyc, cyc0, winst * 64 / ms / 1e6);
    return 0;
}

// ------------------------------------------------------------------------------------------- unaligned correctness
__global__ void lds_unaligned_check(uint32_t *bad, uint32_t stride) {
    __shared__ __attribute__((aligned(16))) unsigned char s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = static_cast<unsigned char>(i * 37 + (i >> 8));
    __syncthreads();
    uint32_t mism = 0;
    for (uint32_t e = threadIdx.x; e * stride + 8 <= 8192; e += blockDim.x) {
        uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s)) + e * stride;
        uint64_t q;
        asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
        uint64_t want = 0;
        for (int b = 7; b >= 0; --b) want = (want << 8) | s[e * stride + b];
        mism += q != want;
        uint32_t d;
        asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory");
        mism += d != static_cast<uint32_t>(want);
        asm volatile("ds_read_u16 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory");
        mism += d != static_cast<uint32_t>(want & 0xffff);
    }
    atomicAdd(bad, mism);
}

// ------------------------------------------------------------------------------------------- LDS stores by active lanes
template <int WIDTH>
__global__ __launch_bounds__(1024) void lds_write_kernel(uint32_t *out, uint64_t lane_mask, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63;
    uint32_t addr = (threadIdx.x >> 6) * 2048 + lane * 8;
    const bool on = (lane_mask >> lane) & 1ull;
    uint32_t v = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (WIDTH == 8) { uint64_t q = v; asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(q)); }
                if (WIDTH == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v));
                if (WIDTH == 2) asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (reinterpret_cast<uint32_t *>(smem)[threadIdx.x] == 0x12345u && v == 77) out[0] = 1;
}

template <int WIDTH>
static int run_lds_write(const char *name, uint64_t mask, uint32_t *out) {
    const int iters = 512, blocks = g_cus, threads = 1024;
    double ms = 0;
    if (time_kernel([&] { hipLaunchKernelGGL(lds_write_kernel<WIDTH>, dim3(blocks), dim3(threads), 64 * 1024, 0, out, mask, iters); }, &ms)) return 1;
    const double winst = double(blocks) * threads / 64 * iters * 8;
    printf("LDSW %-44s %6.2f CU-cycles per wave-store\n", name, (ms * 1e-3) * g_clk / (winst / g_cus));
    return 0;
}

// ------------------------------------------------------------------------------------------- cross-lane through the LDS crossbar
__global__ __launch_bounds__(1024) void bpermute_kernel(uint32_t *out, int iters) {
    uint32_t v = threadIdx.x * 2654435761u, a = ((threadIdx.x * 7u) & 63u) * 4u, acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t r;
            asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(v));
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            acc += r;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345u) out[0] = acc;
}

// ------------------------------------------------------------------------------------------- VALU and LDS together
// 2 random b32 lookups + N VALU per iteration: do the pipes overlap?
template <int NVALU>
__global__ __launch_bounds__(1024) void mixed_kernel(uint32_t *out, uint32_t span_entries, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0, y = x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t addr = __umulhi(x, span_entries) * 4u, v0;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(addr));
#pragma unroll
            for (int k = 0; k < NVALU; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(x));
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            acc += v0;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc + y == 0x12345u) out[0] = acc;
}
template <int NVALU>
static int run_mixed(uint32_t *out) {
    const int iters = 512, blocks = g_cus, threads = 1024;
    const size_t lds = 160 * 1024 - 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mixed_kernel<NVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    double ms = 0;
    if (time_kernel([&] { hipLaunchKernelGGL(mixed_kernel<NVALU>, dim3(blocks), dim3(threads), lds, 0, out, 19683u, iters); }, &ms)) return 1;
    const double winst = double(blocks) * threads / 64 * iters * 8;
    printf("MIX  1 random b32 lookup + %2d VALU (+3 address)     %6.2f CU-cycles per lookup group\n", NVALU, (ms * 1e-3) * g_clk / (winst / g_cus));
    return 0;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz (rates below assume 2.4 GHz)\n", prop.gcnArchName, g_cus, prop.clockRate / 1000);
    const bool only_unaligned = argc > 1 && !strcmp(argv[1], "unaligned");
    uint32_t *out;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(out, 0, 64));
    if (only_unaligned) {
        for (uint32_t stride : {8u, 4u, 6u, 5u, 7u, 3u, 1u}) {
            CHECK(hipMemset(out, 0, 4));
            hipLaunchKernelGGL(lds_unaligned_check, dim3(1), dim3(256), 0, 0, out, stride);
            hipError_t e = hipDeviceSynchronize();
            uint32_t bad = 0;
            if (e == hipSuccess) CHECK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
            printf("UNALIGNED ds_read_b64/b32/u16 at stride %u: %s, %u mismatches\n", stride, hipGetErrorString(e), bad);
            if (e != hipSuccess) return 2;
        }
        return 0;
    }
#define RV(fn) if (run_valu(#fn, fn)) return 1;
    RV(k_add) RV(k_and) RV(k_lshl) RV(k_bfe) RV(k_bfev) RV(k_lshlor) RV(k_mad24) RV(k_add3) RV(k_alignbit) RV(k_perm) RV(k_bcnt) RV(k_ffbl)
    RV(k_mov) RV(k_movsdwa) RV(k_addsdwa) RV(k_lshlsdwa) RV(k_movdpp) RV(k_movwshr) RV(k_adddpp) RV(k_fma) RV(k_fadd) RV(k_fmul) RV(k_fmac)
    RV(k_cvtub0) RV(k_cvtub2) RV(k_cvtu32) RV(k_cvtf32) RV(k_pkmad16) RV(k_pkadd16) RV(k_pklshl16) RV(k_lshladd) RV(k_cndmask) RV(k_cmp) RV(k_cmps)
    RV(k_mbcnt) RV(k_dot4) RV(k_sad) RV(k_mullo) RV(k_maxu) RV(k_xad) RV(k_andor) RV(k_or3) RV(k_bfi) RV(k_readlane) RV(k_madu16)
    RV(k_sub) RV(k_or) RV(k_xor) RV(k_minu) RV(k_pksub16) RV(k_pkmin16) RV(k_madu32u16) RV(k_dot2u16) RV(k_lshrsdwa) RV(k_lshrv) RV(k_subsdwa) RV(k_andk)
    RV(k_pkfma) RV(k_lshladd64) RV(k_mad64) RV(k_cndmask_s)
    if (run_valu("k_cmp_cnd (2 instr)", k_cmp_cnd, 16)) return 1;
    if (run_valu("k_cmps_cnd (2 instr)", k_cmps_cnd, 16)) return 1;

    // LDS reads
    if (run_lds<0>("ds_read_u8  class table, 27 symbols (a-z, space)", 0, 1, 27, out)) return 1;
    if (run_lds<0>("ds_read_u8  class table, 95 symbols", 0, 1, 95, out)) return 1;
    if (run_lds<0>("ds_read_u8  class table, 256 symbols", 0, 1, 256, out)) return 1;
    if (run_lds<2>("ds_read_b32 u32 table, 27 symbols", 0, 4, 27, out)) return 1;
    if (run_lds<2>("ds_read_b32 u32 table, 95 symbols", 0, 4, 95, out)) return 1;
    if (run_lds<2>("ds_read_b32 random over 19683 entries (79 KB)", 19683, 4, 0, out)) return 1;
    if (run_lds<2>("ds_read_b32 random over 729 entries", 729, 4, 0, out)) return 1;
    if (run_lds<1>("ds_read_u16 random over 19683 entries (39 KB)", 19683, 2, 0, out)) return 1;
    if (run_lds<3>("ds_read_b64 random aligned, 19683 entries (157 KB)", 19683, 8, 0, out)) return 1;
    if (run_lds<3>("ds_read_b64 random aligned, 2578 entries (21 KB)", 2578, 8, 0, out)) return 1;
    if (run_lds<5>("ds_read_b64 random at 6-byte stride (118 KB)", 19683, 6, 0, out)) return 1;
    if (run_lds<6>("ds_read_b64 random at 5-byte stride (98 KB)", 19683, 5, 0, out)) return 1;
    if (run_lds<4>("ds_read_b128 random aligned, 9840 entries (157 KB)", 9840, 16, 0, out)) return 1;
    if (run_lds<7>("ds_read2_b32 random (two adjacent dwords), 79 KB", 19683, 4, 0, out)) return 1;
    // LDS stores
    if (run_lds_write<8>("ds_write_b64, 64 lanes", ~0ull, out)) return 1;
    if (run_lds_write<8>("ds_write_b64, 7 lanes", 0x0102040810204080ull >> 1, out)) return 1;
    if (run_lds_write<8>("ds_write_b64, 1 lane", 1ull << 17, out)) return 1;
    if (run_lds_write<4>("ds_write_b32, 64 lanes", ~0ull, out)) return 1;
    if (run_lds_write<4>("ds_write_b32, 7 lanes", 0x0102040810204080ull >> 1, out)) return 1;
    if (run_lds_write<2>("ds_write_b16, 64 lanes", ~0ull, out)) return 1;
    if (run_lds_write<2>("ds_write_b16, 7 lanes", 0x0102040810204080ull >> 1, out)) return 1;
    {
        double ms = 0;
        const int iters = 512;
        if (time_kernel([&] { hipLaunchKernelGGL(bpermute_kernel, dim3(g_cus), dim3(1024), 0, 0, out, iters); }, &ms)) return 1;
        const double winst = double(g_cus) * 16 * iters * 8;
        printf("XLN  ds_bpermute_b32                                  %6.2f CU-cycles per wave-instr\n", (ms * 1e-3) * g_clk / (winst / g_cus));
    }
    if (run_mixed<0>(out)) return 1;
    if (run_mixed<4>(out)) return 1;
    if (run_mixed<8>(out)) return 1;
    if (run_mixed<16>(out)) return 1;
    if (run_mixed<24>(out)) return 1;
    return 0;
}
