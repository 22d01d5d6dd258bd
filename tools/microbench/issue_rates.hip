// issue_rates.hip — instruction issue cost on gfx950 (MI355X) for the instruction mix of dg_raster_tiles.
//
// For every instruction (or short pattern) below: a workgroup of 256 threads (one wave per SIMD of its CU) runs a loop of
// REP x UNROLL independent copies; k workgroups per CU (k = 1, 2, 4, 8 waves per SIMD) run together on every CU.  The
// kernel stamps s_memtime around the loop; the table gives
//       cycles per instruction per SIMD  =  median(wave cycles) / (instructions per wave x waves per SIMD)
// where "wave cycles" is taken three ways: the median wave's own loop time ("med": under oldest-first arbitration the old
// waves finish early, so this under-states the cost), the slowest wave's loop time ("max") and the span from the first
// loop start to the last loop end on the whole chip divided the same way ("all": includes dispatch skew).  max / all are
// i.e. the issue cost the SIMD (or, for scalar work, the CU's scalar unit shared by 4 SIMDs) pays per wave-instruction when
// k waves compete for it.  This is the number an "issue-bound" roofline has to be priced with.
//
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o issue_rates tools/microbench/issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REP = 256;      // loop iterations
constexpr int UNROLL = 32;    // instruction copies per iteration

// Eight independent destination registers; sources are loop-invariant so there is no dependent chain.
#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define BODY32(op) R8(op) R8(op) R8(op) R8(op)

#define KERNEL(name, ASMBODY, CLOBBER...)                                                                                  \
    __global__ __launch_bounds__(256) void k_##name(unsigned long long *out, const float *in) {                            \
        float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)], c = in[128 + (threadIdx.x & 63)];                 \
        unsigned ai = __float_as_uint(a) | 1u, bi = (__float_as_uint(b) & 31u) | 1u;                                       \
        float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;                                                \
        __shared__ unsigned lds[1024];                                                                                     \
        lds[threadIdx.x] = ai; lds[threadIdx.x + 256] = bi; lds[threadIdx.x + 512] = ai; lds[threadIdx.x + 768] = bi;      \
        __syncthreads();                                                                                                   \
        unsigned laddr = (threadIdx.x & 63) * 4; unsigned lb = 0;                                                          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int i = 0; i < REP; i++) {                                                                                    \
            asm volatile(ASMBODY                                                                                           \
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)                  \
                         : "v"(a), "v"(b), "v"(c), "v"(ai), "v"(bi), "v"(laddr), "v"(lb)                                  \
                         : "memory", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "scc", ##CLOBBER);              \
        }                                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                              \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = t1; } \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123.456f) out[0] = 0;                                                 \
    }

// %0..%7 = d0..d7, %8 = a, %9 = b, %10 = c, %11 = ai, %12 = bi, %13 = laddr, %14 = lb
#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %10\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %8, %9\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %8, %9\n"
#define OP_MAX(i) "v_max_f32 %" #i ", %8, %9\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %11, %12\n"
#define OP_AND(i) "v_and_b32 %" #i ", %11, %12\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", %12, %11\n"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %11, %12, %11\n"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %11, %12, %11\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %11, 3, 6\n"
#define OP_BFI(i) "v_bfi_b32 %" #i ", %12, %11, %12\n"
#define OP_MED3(i) "v_med3_i32 %" #i ", %11, %12, %11\n"
#define OP_CVTI(i) "v_cvt_i32_f32 %" #i ", %8\n"
#define OP_CVTF(i) "v_cvt_f32_i32 %" #i ", %11\n"
#define OP_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %11\n"
#define OP_TRUNC(i) "v_trunc_f32 %" #i ", %8\n"
#define OP_PKU8(i) "v_cvt_pk_u8_f32 %" #i ", %8, 1, %11\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %11, %12, vcc\n"
#define OP_CMP(i) "v_cmp_lt_u32 vcc, %11, %12\n"
#define OP_CMPS(i) "v_cmp_lt_u32 s[40:41], %11, %12\n"
#define OP_CMPX(i) "v_cmp_le_u32 vcc, %11, %12\n v_cndmask_b32 %" #i ", %11, %12, vcc\n"
#define OP_CNDS(i) "v_cndmask_b32 %" #i ", %11, %12, s[44:45]\n"
#define OP_CNDMIX(i) "v_cndmask_b32 %" #i ", %11, %12, vcc\n v_add_u32 %" #i ", %11, %12\n"
#define OP_MIXHF(i) "v_fma_f32 %" #i ", %8, %9, %10\n v_cvt_i32_f32 %" #i ", %8\n"
#define OP_MIXHS(i) "v_cvt_i32_f32 %" #i ", %8\n s_add_u32 s4" #i ", s4" #i ", 3\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %11\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %8\n"
#define OP_FIXUP(i) "v_div_fixup_f32 %" #i ", %8, %9, %10\n"
#define OP_MADU24(i) "v_mad_u32_u24 %" #i ", %11, %12, %11\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %11, %12\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %11, %12, %11\n"
#define OP_PKADD16(i) "v_pk_add_u16 %" #i ", %11, %12\n"
#define OP_CVTPKI16(i) "v_cvt_pk_i16_i32 %" #i ", %11, %12\n"
#define OP_SDWA_CVTU(i) "v_cvt_u32_f32_sdwa %" #i ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
#define OP_SDWA_MUL(i) "v_mul_f32_sdwa %" #i ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define OP_DPP_MOV(i) "v_mov_b32_dpp %" #i ", %11 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_READLANE(i) "v_readlane_b32 s4" #i ", %11, 5\n"
#define OP_READFIRST(i) "v_readfirstlane_b32 s4" #i ", %11\n"
#define OP_SADD(i) "s_add_u32 s4" #i ", s4" #i ", 3\n"
#define OP_SAND64(i) "s_and_b64 s[42:43], s[44:45], s[46:47]\n"
#define OP_SFF1(i) "s_ff1_i32_b64 s4" #i ", s[46:47]\n"
#define OP_SBFE(i) "s_bfe_u32 s4" #i ", s47, 0x60003\n"
#define OP_SCMP(i) "s_cmp_lt_u32 s4" #i ", s47\n"
#define OP_SNOP(i) "s_nop 0\n"
// one VALU + one independent SALU from the SAME wave (does a wave overlap them?)
#define OP_VS(i) "v_fma_f32 %" #i ", %8, %9, %10\n s_add_u32 s4" #i ", s4" #i ", 3\n"
#define OP_VSS(i) "v_fma_f32 %" #i ", %8, %9, %10\n s_add_u32 s4" #i ", s4" #i ", 3\n s_and_b32 s48, s47, s46\n"
// never-taken conditional branch (scc = 0 after s_cmp_eq of different values) and an always-taken one
#define OP_BR_NT(i) "s_cmp_eq_u32 s47, -1\n s_cbranch_scc1 9f\n"
#define OP_BR_T(i) "s_branch 1" #i "f\n s_nop 0\n1" #i ":\n"
#define OP_BRV_NT(i) "s_cbranch_vccz 9f\n"
#define OP_BREXECZ(i) "s_cbranch_execz 9f\n"
// the scalar skeleton of the span walk: ctz + clear-lowest + readlane + branch on a flag
#define OP_WALK(i) "s_ff1_i32_b64 s48, s[46:47]\n s_lshl_b32 s49, 1, s48\n s_andn2_b32 s46, s46, s49\n v_readlane_b32 s4" #i ", %11, 5\n s_bitcmp1_b32 s4" #i ", 15\n s_cbranch_scc1 9f\n"
// LDS
#define OP_LDSR32(i) "ds_read_b32 %" #i ", %13\n"
#define OP_LDSR128B(i) "ds_read_b128 v[20:23], %14\n"   /* broadcast: same address in all lanes */
#define OP_LDSW32(i) "ds_write_b32 %13, %11\n"


// ---- round 5 additions: which instructions share the "simple" class (two per 4.25 clocks and SIMD), packed f32, mixes ----
#define OP_SUBU(i) "v_sub_u32 %" #i ", %11, %12\n"
#define OP_SUBREV(i) "v_subrev_u32 %" #i ", %11, %12\n"
#define OP_OR(i) "v_or_b32 %" #i ", %11, %12\n"
#define OP_XOR(i) "v_xor_b32 %" #i ", %11, %12\n"
#define OP_SUBF(i) "v_sub_f32 %" #i ", %8, %9\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %11, 4, %12\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %11, %12, %11\n"
#define OP_ASHR(i) "v_ashrrev_i32 %" #i ", 16, %11\n"
#define OP_LSHR(i) "v_lshrrev_b32 %" #i ", 16, %11\n"
#define OP_MAXU(i) "v_max_u32 %" #i ", %11, %12\n"
#define OP_MINF(i) "v_min_f32 %" #i ", %8, %9\n"
#define OP_MULU24(i) "v_mul_u32_u24 %" #i ", %11, %12\n"
#define OP_MADI24(i) "v_mad_i32_i24 %" #i ", %11, %12, %11\n"
#define OP_FLOOR(i) "v_floor_f32 %" #i ", %8\n"
#define OP_CVTU(i) "v_cvt_u32_f32 %" #i ", %8\n"
#define OP_CVTFU(i) "v_cvt_f32_u32 %" #i ", %11\n"
#define OP_CVTUB0(i) "v_cvt_f32_ubyte0 %" #i ", %11\n"
#define OP_SDWA_SUB(i) "v_sub_u32_sdwa %" #i ", %11, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define OP_SDWA_ADD(i) "v_add_u32_sdwa %" #i ", %11, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_SDWA_CVTF(i) "v_cvt_f32_i32_sdwa %" #i ", sext(%11) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
#define OP_CMPF(i) "v_cmp_gt_f32 vcc, %8, %9\n"
#define OP_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %11, %12\n"
#define OP_MOVS(i) "v_mov_b32 %" #i ", s46\n"
#define OP_ADDS(i) "v_add_u32 %" #i ", s46, %11\n"
#define OP_MULS(i) "v_mul_f32 %" #i ", s46, %8\n"
#define OP_FMAS(i) "v_fma_f32 %" #i ", %8, s46, %9\n"
#define OP_ADDLIT(i) "v_add_u32 %" #i ", 0x12345, %11\n"
#define OP_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", -1, 0\n"
#define OP_BPERM(i) "ds_bpermute_b32 %" #i ", %13, %11\n"
// packed f32 (register pairs named explicitly)
#define P8(op) op("20:21") op("22:23") op("24:25") op("26:27") op("28:29") op("30:31") op("32:33") op("34:35")
#define BODY32P(op) P8(op) P8(op) P8(op) P8(op)
#define PKMUL(d) "v_pk_mul_f32 v[" d "], v[40:41], v[42:43]\n"
#define PKFMA(d) "v_pk_fma_f32 v[" d "], v[40:41], v[42:43], v[44:45]\n"
#define PKADD(d) "v_pk_add_f32 v[" d "], v[40:41], v[42:43]\n"
#define PKMULCVT(d) "v_pk_mul_f32 v[" d "], v[40:41], v[42:43]\n v_cvt_i32_f32 v46, v40\n"
#define PKMULFMA(d) "v_pk_mul_f32 v[" d "], v[40:41], v[42:43]\n v_fma_f32 v46, v40, v41, v42\n"
#define PKCLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42", "v43", "v44", "v45", "v46"
// mixes: how many simple instructions ride along with one complex one?
#define OP_MIX21(i) "v_fma_f32 %" #i ", %8, %9, %10\n v_mul_f32 %" #i ", %8, %9\n v_cvt_i32_f32 %" #i ", %8\n"
#define OP_MIX12(i) "v_fma_f32 %" #i ", %8, %9, %10\n v_cvt_i32_f32 %" #i ", %8\n v_med3_i32 %" #i ", %11, %12, %11\n"
#define OP_MIXRL(i) "v_readlane_b32 s4" #i ", %11, 5\n v_fma_f32 %" #i ", %8, %9, %10\n"
#define OP_MIXCVCV(i) "v_cvt_i32_f32 %" #i ", %8\n v_cndmask_b32 %" #i ", %11, %12, s[44:45]\n"
#define OP_MIXLDS(i) "ds_read_b32 %" #i ", %13\n v_cvt_i32_f32 v46, %8\n v_fma_f32 v47, %8, %9, %10\n"
#define OP_SNOP1(i) "v_fma_f32 %" #i ", %8, %9, %10\n s_nop 1\n"

KERNEL(fma, BODY32(OP_FMA))
KERNEL(mul, BODY32(OP_MUL))
KERNEL(add, BODY32(OP_ADD))
KERNEL(max, BODY32(OP_MAX))
KERNEL(addu, BODY32(OP_ADDU))
KERNEL(and, BODY32(OP_AND))
KERNEL(lshl, BODY32(OP_LSHL))
KERNEL(lshlor, BODY32(OP_LSHLOR))
KERNEL(andor, BODY32(OP_ANDOR))
KERNEL(bfe, BODY32(OP_BFE))
KERNEL(bfi, BODY32(OP_BFI))
KERNEL(med3, BODY32(OP_MED3))
KERNEL(cvti, BODY32(OP_CVTI))
KERNEL(cvtf, BODY32(OP_CVTF))
KERNEL(cvtub, BODY32(OP_CVTUB))
KERNEL(trunc, BODY32(OP_TRUNC))
KERNEL(pku8, BODY32(OP_PKU8))
KERNEL(cndmask, BODY32(OP_CNDMASK))
KERNEL(cmp, BODY32(OP_CMP))
KERNEL(cmps, BODY32(OP_CMPS))
KERNEL(cmpx, BODY32(OP_CMPX))
KERNEL(cnds, BODY32(OP_CNDS))
KERNEL(cndmix, BODY32(OP_CNDMIX))
KERNEL(mixhf, BODY32(OP_MIXHF))
KERNEL(mixhs, BODY32(OP_MIXHS))
KERNEL(mov, BODY32(OP_MOV))
KERNEL(rcp, BODY32(OP_RCP))
KERNEL(fixup, BODY32(OP_FIXUP))
KERNEL(madu24, BODY32(OP_MADU24))
KERNEL(mullo, BODY32(OP_MULLO))
KERNEL(perm, BODY32(OP_PERM))
KERNEL(pkadd16, BODY32(OP_PKADD16))
KERNEL(cvtpki16, BODY32(OP_CVTPKI16))
KERNEL(sdwa_cvtu, BODY32(OP_SDWA_CVTU))
KERNEL(sdwa_mul, BODY32(OP_SDWA_MUL))
KERNEL(dpp_mov, BODY32(OP_DPP_MOV))
KERNEL(readlane, BODY32(OP_READLANE))
KERNEL(readfirst, BODY32(OP_READFIRST))
KERNEL(sadd, BODY32(OP_SADD))
KERNEL(sand64, BODY32(OP_SAND64))
KERNEL(sff1, BODY32(OP_SFF1))
KERNEL(sbfe, BODY32(OP_SBFE))
KERNEL(scmp, BODY32(OP_SCMP))
KERNEL(snop, BODY32(OP_SNOP))
KERNEL(vs, BODY32(OP_VS))
KERNEL(vss, BODY32(OP_VSS))
KERNEL(br_nt, BODY32(OP_BR_NT) "9:\n")
KERNEL(br_t, BODY32(OP_BR_T))
KERNEL(brv_nt, "v_cmp_lt_u32 vcc, %11, %11\n" BODY32(OP_BRV_NT) "9:\n")
KERNEL(brexecz, BODY32(OP_BREXECZ) "9:\n")
KERNEL(walk, "s_mov_b32 s46, -1\n s_mov_b32 s47, -1\n" BODY32(OP_WALK) "9:\n")
KERNEL(ldsr32, BODY32(OP_LDSR32) "s_waitcnt lgkmcnt(0)\n")
KERNEL(ldsr128b, BODY32(OP_LDSR128B) "s_waitcnt lgkmcnt(0)\n", "v20", "v21", "v22", "v23")
KERNEL(ldsw32, BODY32(OP_LDSW32) "s_waitcnt lgkmcnt(0)\n")

KERNEL(subu, BODY32(OP_SUBU))
KERNEL(subrev, BODY32(OP_SUBREV))
KERNEL(or, BODY32(OP_OR))
KERNEL(xor, BODY32(OP_XOR))
KERNEL(subf, BODY32(OP_SUBF))
KERNEL(fmac, BODY32(OP_FMAC))
KERNEL(lshladd, BODY32(OP_LSHLADD))
KERNEL(add3, BODY32(OP_ADD3))
KERNEL(ashr, BODY32(OP_ASHR))
KERNEL(lshr, BODY32(OP_LSHR))
KERNEL(maxu, BODY32(OP_MAXU))
KERNEL(minf, BODY32(OP_MINF))
KERNEL(mulu24, BODY32(OP_MULU24))
KERNEL(madi24, BODY32(OP_MADI24))
KERNEL(floor, BODY32(OP_FLOOR))
KERNEL(cvtu, BODY32(OP_CVTU))
KERNEL(cvtfu, BODY32(OP_CVTFU))
KERNEL(cvtub0, BODY32(OP_CVTUB0))
KERNEL(sdwa_sub, BODY32(OP_SDWA_SUB))
KERNEL(sdwa_add, BODY32(OP_SDWA_ADD))
KERNEL(sdwa_cvtf, BODY32(OP_SDWA_CVTF))
KERNEL(cmpf, BODY32(OP_CMPF))
KERNEL(addco, BODY32(OP_ADDCO))
KERNEL(movs, BODY32(OP_MOVS))
KERNEL(adds, BODY32(OP_ADDS))
KERNEL(muls, BODY32(OP_MULS))
KERNEL(fmas, BODY32(OP_FMAS))
KERNEL(addlit, BODY32(OP_ADDLIT))
KERNEL(mbcnt, BODY32(OP_MBCNT))
KERNEL(bperm, BODY32(OP_BPERM) "s_waitcnt lgkmcnt(0)\n")
KERNEL(pkmul, BODY32P(PKMUL), PKCLOB)
KERNEL(pkfma, BODY32P(PKFMA), PKCLOB)
KERNEL(pkadd, BODY32P(PKADD), PKCLOB)
KERNEL(pkmulcvt, BODY32P(PKMULCVT), PKCLOB)
KERNEL(pkmulfma, BODY32P(PKMULFMA), PKCLOB)
KERNEL(mix21, BODY32(OP_MIX21))
KERNEL(mix12, BODY32(OP_MIX12))
KERNEL(mixrl, BODY32(OP_MIXRL))
KERNEL(mixcvcv, BODY32(OP_MIXCVCV))
KERNEL(mixlds, BODY32(OP_MIXLDS) "s_waitcnt lgkmcnt(0)\n", "v46", "v47")
KERNEL(snop1, BODY32(OP_SNOP1))


__global__ __launch_bounds__(256) void k_clock(unsigned long long *out, const float *in) {
    float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)], c = in[128 + (threadIdx.x & 63)];
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 16 * REP; i++)
        asm volatile(BODY32(OP_FMA) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b), "v"(c));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123.456f) out[0] = 0;
}

struct Entry { const char *name; void (*fn)(unsigned long long *, const float *); int per_copy; const char *what; };
#define E(name, n, what) { #name, k_##name, n, what }
static const Entry entries[] = {
    E(fma, 1, "v_fma_f32"), E(mul, 1, "v_mul_f32"), E(add, 1, "v_add_f32"), E(max, 1, "v_max_f32"),
    E(addu, 1, "v_add_u32"), E(and, 1, "v_and_b32"), E(lshl, 1, "v_lshlrev_b32"), E(lshlor, 1, "v_lshl_or_b32"), E(andor, 1, "v_and_or_b32"),
    E(bfe, 1, "v_bfe_u32"), E(bfi, 1, "v_bfi_b32"), E(med3, 1, "v_med3_i32"), E(cvti, 1, "v_cvt_i32_f32"), E(cvtf, 1, "v_cvt_f32_i32"),
    E(cvtub, 1, "v_cvt_f32_ubyte1"), E(trunc, 1, "v_trunc_f32"), E(pku8, 1, "v_cvt_pk_u8_f32"), E(cndmask, 1, "v_cndmask_b32 (vcc)"),
    E(cmp, 1, "v_cmp_lt_u32 vcc"), E(cmps, 1, "v_cmp_lt_u32 sgpr pair"), E(cmpx, 2, "v_cmp -> v_cndmask (dependent through vcc), per instruction"),
    E(cnds, 1, "v_cndmask_b32 (sgpr pair)"), E(cndmix, 2, "v_cndmask vcc + v_add_u32 alternating, per instr"),
    E(mixhf, 2, "v_fma_f32 + v_cvt_i32_f32 alternating, per instr"), E(mixhs, 2, "v_cvt_i32_f32 + s_add_u32 alternating, per instr"),
    E(mov, 1, "v_mov_b32"), E(rcp, 1, "v_rcp_f32"), E(fixup, 1, "v_div_fixup_f32"), E(madu24, 1, "v_mad_u32_u24"), E(mullo, 1, "v_mul_lo_u32"),
    E(perm, 1, "v_perm_b32"), E(pkadd16, 1, "v_pk_add_u16"), E(cvtpki16, 1, "v_cvt_pk_i16_i32"), E(sdwa_cvtu, 1, "v_cvt_u32_f32_sdwa dst_sel:BYTE_1 preserve"),
    E(sdwa_mul, 1, "v_mul_f32_sdwa"), E(dpp_mov, 1, "v_mov_b32_dpp row_shr:1"), E(readlane, 1, "v_readlane_b32"), E(readfirst, 1, "v_readfirstlane_b32"),
    E(sadd, 1, "s_add_u32"), E(sand64, 1, "s_and_b64"), E(sff1, 1, "s_ff1_i32_b64"), E(sbfe, 1, "s_bfe_u32"), E(scmp, 1, "s_cmp_lt_u32"), E(snop, 1, "s_nop 0"),
    E(vs, 2, "v_fma_f32 + s_add_u32 alternating in one wave, per instruction"), E(vss, 3, "v_fma_f32 + 2 SALU alternating in one wave, per instruction"),
    E(br_nt, 2, "s_cmp + s_cbranch_scc1 never taken, per instruction"), E(br_t, 2, "s_branch taken (+ skipped s_nop), per taken branch pair"),
    E(brv_nt, 1, "s_cbranch_vccz never taken"), E(brexecz, 1, "s_cbranch_execz never taken"),
    E(walk, 6, "span-walk skeleton: ff1, lshl, andn2, v_readlane, bitcmp, cbranch (not taken), per instruction"),
    E(subu, 1, "v_sub_u32"), E(subrev, 1, "v_subrev_u32"), E(or, 1, "v_or_b32"), E(xor, 1, "v_xor_b32"), E(subf, 1, "v_sub_f32"), E(fmac, 1, "v_fmac_f32"),
    E(lshladd, 1, "v_lshl_add_u32"), E(add3, 1, "v_add3_u32"), E(ashr, 1, "v_ashrrev_i32"), E(lshr, 1, "v_lshrrev_b32"), E(maxu, 1, "v_max_u32"), E(minf, 1, "v_min_f32"),
    E(mulu24, 1, "v_mul_u32_u24"), E(madi24, 1, "v_mad_i32_i24"), E(floor, 1, "v_floor_f32"), E(cvtu, 1, "v_cvt_u32_f32"), E(cvtfu, 1, "v_cvt_f32_u32"), E(cvtub0, 1, "v_cvt_f32_ubyte0"),
    E(sdwa_sub, 1, "v_sub_u32_sdwa src1 WORD_0"), E(sdwa_add, 1, "v_add_u32_sdwa src0 WORD_1"), E(sdwa_cvtf, 1, "v_cvt_f32_i32_sdwa sext WORD_0"), E(cmpf, 1, "v_cmp_gt_f32 vcc"),
    E(addco, 1, "v_add_co_u32 vcc"), E(movs, 1, "v_mov_b32 v, s"), E(adds, 1, "v_add_u32 v, s, v"), E(muls, 1, "v_mul_f32 v, s, v"), E(fmas, 1, "v_fma_f32 v, v, s, v"),
    E(addlit, 1, "v_add_u32 v, literal, v"), E(mbcnt, 1, "v_mbcnt_lo_u32_b32"), E(bperm, 1, "ds_bpermute_b32"),
    E(pkmul, 1, "v_pk_mul_f32 (per instruction = 2 multiplies per lane)"), E(pkfma, 1, "v_pk_fma_f32"), E(pkadd, 1, "v_pk_add_f32"),
    E(pkmulcvt, 2, "v_pk_mul_f32 + v_cvt_i32_f32 alternating, per instr"), E(pkmulfma, 2, "v_pk_mul_f32 + v_fma_f32 alternating, per instr"),
    E(mix21, 3, "v_fma + v_mul + v_cvt_i32_f32 (2 simple : 1 complex), per instr"), E(mix12, 3, "v_fma + v_cvt + v_med3 (1 simple : 2 complex), per instr"),
    E(mixrl, 2, "v_readlane + v_fma alternating, per instr"), E(mixcvcv, 2, "v_cvt + v_cndmask(sgpr) alternating (2 complex), per instr"),
    E(mixlds, 3, "ds_read_b32 + v_cvt + v_fma, per instr"), E(snop1, 2, "v_fma + s_nop 1, per instr"),
    E(ldsr32, 1, "ds_read_b32 (lane-linear)"), E(ldsr128b, 1, "ds_read_b128 broadcast address"), E(ldsw32, 1, "ds_write_b32 (lane-linear)"),
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float h_in[192];
    for (int i = 0; i < 192; i++) h_in[i] = 1.0f + 0.001f * (float)i;
    float *d_in; unsigned long long *d_out;
    CHECK(hipMalloc(&d_in, sizeof(h_in)));
    CHECK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&d_out, (size_t)max_blocks * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_out((size_t)max_blocks * 8);
    // (s_memtime is not consistent across XCDs, so a "first start .. last end over all waves" figure is meaningless and is not printed)
    printf("%-10s %-60s | %-27s | %-27s\n", "", "cycles per wave-instruction per SIMD, k waves per SIMD:", "slowest wave (max)", "median wave (med)");
    printf("%-10s %-60s | %6s %6s %6s %6s | %6s %6s %6s %6s\n", "name", "instruction", "k=1", "k=2", "k=4", "k=8", "k=1", "k=2", "k=4", "k=8");
    for (const Entry &e : entries) {
        double r[3][4];
        int ki = 0;
        for (int k : {1, 2, 4, 8}) {
            const int blocks = cus * k;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_in);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h_out.data(), d_out, (size_t)blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v;
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int w = 0; w < blocks * 4; w++) {
                v.push_back(h_out[2 * w + 1] - h_out[2 * w]);
                tmin = std::min(tmin, h_out[2 * w]); tmax = std::max(tmax, h_out[2 * w + 1]);
            }
            std::sort(v.begin(), v.end());
            const double n_inst = (double)REP * UNROLL * e.per_copy * k;
            r[0][ki] = (double)v.back() / n_inst; r[1][ki] = (double)(tmax - tmin) / n_inst; r[2][ki] = (double)v[v.size() / 2] / n_inst;
            ki++;
        }
        printf("%-10s %-60s |", e.name, e.what);
        for (int m = 0; m < 3; m += 2) { for (int j = 0; j < 4; j++) printf(" %6.2f", r[m][j]); printf(" %s", m < 2 ? "|" : "\n"); }
        fflush(stdout);
    }
    // s_memtime tick vs the 100 MHz constant clock (s_memrealtime) and vs wall time
    hipLaunchKernelGGL(k_clock, dim3(cus * 8), dim3(256), 0, 0, d_out, d_in);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_out.data(), d_out, (size_t)cus * 8 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < cus * 8 * 4; w++) ghz.push_back((double)h_out[2 * w] / (double)h_out[2 * w + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("clock: s_memtime ticks per s_memrealtime tick x 100 MHz over a v_fma loop at k=8: min %.3f median %.3f max %.3f GHz\n", ghz.front(), ghz[ghz.size() / 2], ghz.back());
    return 0;
}
