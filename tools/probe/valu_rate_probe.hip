// Issue cost of the instructions an exp2 could be built from, on one SIMD of gfx950 (can part of the
// softmax's v_exp_f32 work move to packed-f32 polynomial arithmetic?).
// Each kernel runs REP x 64 independent instructions of one kind (or an interleaved mix) in a wave and reports shader
// clock cycles per instruction; "2 waves" puts two waves on the same SIMD (a 512-thread block: waves 0 and 4 share SIMD 0)
// with a DIFFERENT instruction kind each, which shows whether two kinds overlap across waves.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/valu_rate_probe.hip -o tools/probe/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 256
// 16 destination registers per kind, sources never written: no dependency stalls, no bank conflicts worth naming
#define X16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define EXP(n)   "v_exp_f32 v[" #n "+8], v1\n"
#define LOG(n)   "v_log_f32 v[" #n "+8], v1\n"
#define RCP(n)   "v_rcp_f32 v[" #n "+8], v1\n"
#define FMA(n)   "v_fma_f32 v[" #n "+8], v1, v2, v3\n"
#define RND(n)   "v_rndne_f32 v[" #n "+8], v1\n"
#define LDEXP(n) "v_ldexp_f32 v[" #n "+8], v1, v4\n"
#define LSHLADD(n) "v_lshl_add_u32 v[" #n "+8], v4, 23, v1\n"
#define CVTI(n)  "v_cvt_i32_f32 v[" #n "+8], v1\n"
#define PKFMA(n) "v_pk_fma_f32 v[2*" #n "+24:2*" #n "+25], v[2:3], v[4:5], v[6:7]\n"
#define PKADD(n) "v_pk_add_f32 v[2*" #n "+24:2*" #n "+25], v[2:3], v[4:5]\n"
#define PKMUL(n) "v_pk_mul_f32 v[2*" #n "+24:2*" #n "+25], v[2:3], v[4:5]\n"
#define MIX_EXP_PKFMA(n) EXP(n) PKFMA(n)
#define MIX_EXP_FMA(n)   EXP(n) FMA(n)
#define MIX_EXP_3PK(n)   EXP(n) PKFMA(n) PKADD(n) PKMUL(n)

#define CLOBBERS "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
                 "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
                 "v48","v49","v50","v51","v52","v53","v54","v55"

#define BODY(NAME, I, PER)                                                                                       \
    __device__ __forceinline__ void body_##NAME() {                                                              \
        for (int r = 0; r < REP; ++r) asm volatile(X16(I) X16(I) X16(I) X16(I) ::: CLOBBERS);                     \
    }                                                                                                            \
    static const int per_##NAME = PER;
BODY(exp, EXP, 1) BODY(log, LOG, 1) BODY(rcp, RCP, 1) BODY(fma, FMA, 1) BODY(rnd, RND, 1) BODY(ldexp, LDEXP, 1)
BODY(lshladd, LSHLADD, 1) BODY(cvti, CVTI, 1) BODY(pkfma, PKFMA, 1) BODY(pkadd, PKADD, 1)
BODY(mix_exp_pkfma, MIX_EXP_PKFMA, 2) BODY(mix_exp_fma, MIX_EXP_FMA, 2) BODY(mix_exp_3pk, MIX_EXP_3PK, 4)

__device__ __forceinline__ void init_regs() {
    asm volatile("v_mov_b32 v1, 0.5\nv_mov_b32 v2, 1.0\nv_mov_b32 v3, 2.0\nv_mov_b32 v4, 1\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 0.5\nv_mov_b32 v7, 2.0" ::: CLOBBERS);
}

// kind index -> body
__device__ __forceinline__ void dispatch(int kind) {
    switch (kind) {
    case 0: body_exp(); break;       case 1: body_log(); break;     case 2: body_rcp(); break;   case 3: body_fma(); break;
    case 4: body_rnd(); break;       case 5: body_ldexp(); break;   case 6: body_lshladd(); break; case 7: body_cvti(); break;
    case 8: body_pkfma(); break;     case 9: body_pkadd(); break;   case 10: body_mix_exp_pkfma(); break;
    case 11: body_mix_exp_fma(); break; case 12: body_mix_exp_3pk(); break;
    }
}

// One block; wave w runs kind (w == 0 ? kindA : kindB) when it is one of the measured waves (0 and, with two, 4: same SIMD).
__global__ void __launch_bounds__(512) probe(long long* out, int kindA, int kindB, int two)
{
    const int wave = threadIdx.x >> 6;
    const bool active = wave == 0 || (two && wave == 4);
    init_regs();
    __syncthreads();
    if (!active) return;
    const int kind = __builtin_amdgcn_readfirstlane(wave == 0 ? kindA : kindB);
    const long long t0 = clock64(), w0 = wall_clock64();
    dispatch(kind);
    const long long t1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) { out[(wave ? 1 : 0) * 2] = t1 - t0; out[(wave ? 1 : 0) * 2 + 1] = w1 - w0; }
}

static const char* names[] = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_fma_f32", "v_rndne_f32", "v_ldexp_f32", "v_lshl_add_u32", "v_cvt_i32_f32",
                              "v_pk_fma_f32", "v_pk_add_f32", "exp+pk_fma interleaved", "exp+fma interleaved", "exp+pk_fma+pk_add+pk_mul"};
static const int per[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 4};

int main()
{
    long long* d; hipMalloc(&d, 64);
    long long h[4];
    const double n = double(REP) * 64;
    printf("one wave on a SIMD, %d x 64 independent instructions; cycles = s_memtime ticks, ns from the 100 MHz wall clock\n", REP);
    for (int k = 0; k < 13; ++k) {
        for (int it = 0; it < 2; ++it) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d, k, k, 0);
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        }
        printf("  %-28s %6.2f ticks / instruction   (%.2f ns; %d instructions per group: %.2f ticks per group)\n", names[k], h[0] / (n * per[k]),
               h[1] * 10.0 / (n * per[k]), per[k], h[0] / n);
    }
    printf("two waves on ONE SIMD, wave A | wave B: ticks per instruction of each while both run\n");
    const int pairs[][2] = {{0, 0}, {8, 8}, {0, 8}, {0, 3}, {3, 3}, {0, 9}};
    for (auto& p : pairs) {
        for (int it = 0; it < 2; ++it) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d, p[0], p[1], 1);
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        }
        printf("  %-14s | %-14s : %6.2f | %6.2f\n", names[p[0]], names[p[1]], h[0] / n, h[2] / n);
    }
    hipFree(d);
    return 0;
}
