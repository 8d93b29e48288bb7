// extern "C" surface of libadk.so (declared in include/adk.h): host-side validation + dispatch.
// Every argument check happens BEFORE a launch: a kernel never sees a shape its grid does not match.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "adk_common.h"

int launch_conv(const adk_op& op, hipStream_t st);
int conv_cotile(int Cout);
int conv_partials_count(const adk_op& op);
int conv_postile(const adk_op& op);
int conv_cotile_used(const adk_op& op);
int conv_gn_check(const adk_op& op, const char** why);
int conv_gn_fusable_here(const adk_op& op);
int launch_gn(const adk_op& op, hipStream_t st, bool apply);
int gn_partials_count(int dtype, int C, int L);
int launch_fir(const adk_op& op, hipStream_t st, bool up);
int launch_attn(const adk_op& op, hipStream_t st);
int launch_pqmf(const adk_op& op, hipStream_t st, bool synth);
int launch_pack_input(const adk_op& op, hipStream_t st);
int launch_unpack_output(const adk_op& op, hipStream_t st);
int launch_ddim_update(const adk_op& op, hipStream_t st);
int launch_step_advance(const adk_op& op, hipStream_t st);
int launch_audio_to_int16(const adk_op& op, hipStream_t st);
int launch_phase_shuffle(const adk_op& op, hipStream_t st, bool to_depth);
int launch_linear(const adk_op& op, hipStream_t st);
int launch_fourier(const adk_op& op, hipStream_t st);

static thread_local char g_err[512] = "";

// ---- scratch of the allocation-free entry point (adk_group_norm_noalloc, bottom of this file) ----------------------
// GroupNorm statistics partials, (sum, sumsq) pairs: [B][adk_partials_count] of ONE call.  Every GPU the library runs
// on holds its own copy of this array at its own address until the process ends (4 MiB).  In host code the array's
// NAME is only a handle for hipGetSymbolAddress: the per-device address it returns (g_scratch_ptr) is what kernels,
// memsets and copies get.
#define ADK_NOALLOC_MAX_PAIRS (512 * 1024)
static __device__ __align__(256) float scratch[ADK_NOALLOC_MAX_PAIRS * 2];
#define ADK_MAX_DEVICES 64
static float* g_scratch_ptr[ADK_MAX_DEVICES];        // per device, filled on the entry point's first call there
static std::mutex g_scratch_mutex;

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define REQUIRE(cond, ...) do { if (!(cond)) return fail(ADK_EINVAL, __VA_ARGS__); } while (0)

static inline int cb_of(int dtype) { return dtype == ADK_F32 ? 16 : 32; }

static int check_geo(const adk_op& o, const char* what)
{
    REQUIRE(o.dtype == ADK_F32 || o.dtype == ADK_BF16 || o.dtype == ADK_F16, "%s: dtype %d unsupported", what, o.dtype);
    REQUIRE(o.B >= 1 && o.L >= 1 && o.P >= 0, "%s: bad geometry B=%d L=%d P=%d", what, o.B, o.L, o.P);
    REQUIRE((long long)o.B * (o.L + 2 * o.P) < (1LL << 25), "%s: B*(L+2P)=%lld rows exceed the 2^25 per-plane limit",
            what, (long long)o.B * (o.L + 2 * o.P));
    return ADK_OK;
}

static int validate(const adk_op& o)
{
    int rc;
    switch (o.kind) {
    case ADK_OP_CONV: {
        if ((rc = check_geo(o, "conv1d"))) return rc;
        const int cb = cb_of(o.dtype);
        REQUIRE(o.taps == 1 || o.taps == 3 || o.taps == 5 || o.taps == 7, "conv1d: taps=%d unsupported (1, 3, 5 or 7)", o.taps);
        const int dil = o.i0 > 1 ? o.i0 : 1;
        REQUIRE(dil == 1 || (o.taps == 7 && (dil == 3 || dil == 9)), "conv1d: dilation %d unsupported (1; 3 or 9 with 7 taps)", dil);
        REQUIRE(o.taps != 5 || (!(o.flags & (ADK_FLAG_CAUSAL | ADK_FLAG_ELU | ADK_FLAG_GROUP_STATS)) && dil == 1),
                "conv1d: causal / ELU / group-statistics convs are built for 1, 3 and 7 taps");
        const int reach = (o.flags & ADK_FLAG_CAUSAL) ? (o.taps - 1) * dil : (o.taps / 2) * dil;
        REQUIRE(o.P >= reach, "conv1d: halo P=%d < %d rows of padding (taps=%d, dilation=%d%s)", o.P, reach, o.taps, dil,
                (o.flags & ADK_FLAG_CAUSAL) ? ", causal" : "");
        REQUIRE(o.Cin >= cb && o.Cin % cb == 0, "conv1d: Cin=%d not a multiple of %d", o.Cin, cb);
        REQUIRE(o.Cout >= 1, "conv1d: Cout=%d", o.Cout);
        const int bn = conv_cotile(o.Cout);
        REQUIRE(o.CoutPad >= o.Cout && o.CoutPad % bn == 0, "conv1d: CoutPad=%d must be a multiple of %d >= Cout=%d",
                o.CoutPad, bn, o.Cout);
        REQUIRE(o.x && o.w && o.y, "conv1d: null tensor");
        REQUIRE(!(o.flags & ADK_FLAG_GROUP_STATS) || !(o.flags & ADK_FLAG_OUT_F32) || o.dtype == ADK_F32,
                "conv1d: group statistics of an f32 output need dtype f32");
        if (o.flags & ADK_FLAG_FUSE_GN) {
            const char* why = "";
            REQUIRE(conv_gn_check(o, &why), "conv1d + GroupNorm fusion: %s", why);
            REQUIRE(o.gamma && o.beta && o.aux0 && o.aux1, "conv1d + GroupNorm fusion: needs gamma, beta, the epoch words (aux0) "
                    "and the granule workspace (aux1)");
            REQUIRE(o.f0 > 0.f, "conv1d + GroupNorm fusion: eps=%g", (double)o.f0);
            REQUIRE(o.i1 >= 0 && o.i1 < 511, "conv1d + GroupNorm fusion: slot %d outside [0, 511)", o.i1);
            REQUIRE(!o.partials_out || o.residual, "conv1d + GroupNorm fusion: output statistics are built for the residual form");
        }
        if (o.flags & ADK_FLAG_PRE_NORM) {
            REQUIRE(o.taps == 1 && !(o.flags & (ADK_FLAG_FUSE_GN | ADK_FLAG_GROUP_STATS | ADK_FLAG_ELU | ADK_FLAG_CAUSAL)),
                    "conv1d: the folded input GroupNorm is built for plain 1x1 convs");
            REQUIRE(o.partials_in && o.n_partials_in >= 1 && o.gamma && o.bias, "conv1d + folded input GroupNorm: needs the "
                    "input's partials, the weight row sums (gamma) and the folded bias");
            REQUIRE(o.i2 >= 1 && o.i2 <= o.Cin && o.f0 > 0.f, "conv1d + folded input GroupNorm: channels i2=%d, eps=%g", o.i2, (double)o.f0);
        }
        if (o.flags & ADK_FLAG_SKIP_CONV) {
            REQUIRE(o.flags & ADK_FLAG_FUSE_GN, "conv1d: the folded skip conv belongs to the conv + GroupNorm fusion");
            REQUIRE(o.L > 128, "conv1d: the folded skip conv is built for the pipelined kernel (L > 128)");
            REQUIRE(o.taps == 3 || o.taps == 5, "conv1d: the folded skip conv is built for 3 and 5 taps");
            REQUIRE(o.residual && o.w2, "conv1d + folded skip conv: needs the block input (residual) and the skip weights (w2)");
            REQUIRE(o.i2 >= cb && o.i2 % cb == 0, "conv1d + folded skip conv: input channels i2=%d not a multiple of %d", o.i2, cb);
        }
        return ADK_OK;
    }
    case ADK_OP_GN_APPLY:
        if ((rc = check_geo(o, "gn_apply"))) return rc;
        REQUIRE(o.Cin >= 1 && o.x && o.y && o.gamma && o.beta, "gn_apply: null tensor / C=%d", o.Cin);
        REQUIRE(o.partials_in && o.n_partials_in >= 1, "gn_apply: needs input partials");
        REQUIRE(o.f0 > 0.f, "gn_apply: eps=%g", (double)o.f0);
        if (o.i1 > 1) {                          // GroupNorm(G, C): groups of whole planes, statistics from GN_STATS
            const int cb = cb_of(o.dtype);
            REQUIRE(o.Cin % o.i1 == 0 && (o.Cin / o.i1) % cb == 0,
                    "gn_apply: %d groups of C=%d are not whole %d-channel planes", o.i1, o.Cin, cb);
            REQUIRE(o.n_partials_in % o.i1 == 0, "gn_apply: %d partials per sample do not split into %d groups "
                    "(grouped statistics come from gn_stats or a conv with ADK_FLAG_GROUP_STATS)", o.n_partials_in, o.i1);
        }
        if (o.flags & ADK_FLAG_ADA) REQUIRE(o.i0 >= o.Cin, "gn_apply: AdaGN row stride %d < C=%d", o.i0, o.Cin);
        return ADK_OK;
    case ADK_OP_GN_STATS:
        if ((rc = check_geo(o, "gn_stats"))) return rc;
        REQUIRE(o.Cin >= 1 && o.x && o.partials_out, "gn_stats: null tensor");
        return ADK_OK;
    case ADK_OP_FIR_DOWN:
    case ADK_OP_FIR_UP: {
        const bool up = o.kind == ADK_OP_FIR_UP;
        if ((rc = check_geo(o, up ? "fir_up2" : "fir_down2"))) return rc;
        REQUIRE(o.taps == 4 || o.taps == 8 || o.taps == 12, "fir: K=%d unsupported (4, 8, 12)", o.taps);
        REQUIRE(o.x && o.y && o.w && o.Cin >= 1, "fir: null tensor");
        if (up) REQUIRE(o.taps / 4 < o.L, "fir_up2: reflect padding %d must be smaller than the input length %d "
                        "(the reference raises the same way)", o.taps / 4, o.L);
        else {
            REQUIRE(o.L % 2 == 0, "fir_down2: L=%d must be even", o.L);
            REQUIRE(o.taps / 2 - 1 < o.L, "fir_down2: reflect padding %d must be smaller than the input length %d "
                    "(the reference raises the same way)", o.taps / 2 - 1, o.L);
        }
        return ADK_OK;
    }
    case ADK_OP_ATTN:
        if ((rc = check_geo(o, "attn1d"))) return rc;
        REQUIRE(o.i0 == 0 || o.i0 == 32 || o.i0 == 64, "attn1d: head dim %d unsupported (32 or 64)", o.i0);
        REQUIRE(o.Cin >= 32 && o.Cin % (o.i0 ? o.i0 : 32) == 0, "attn1d: C=%d must be a multiple of the head dim %d", o.Cin,
                o.i0 ? o.i0 : 32);
        REQUIRE(o.x && o.y, "attn1d: null tensor");
        if (o.flags & ADK_FLAG_QKV) {
            REQUIRE(o.dtype != ADK_F32 && (o.i0 == 0 || o.i0 == 32) && o.L <= 256 && o.P * 4 <= 64,
                    "attn1d + qkv projection: built for 16-bit dtypes, head dim 32, L <= 256 (got dtype %d, L=%d)", o.dtype, o.L);
            REQUIRE(o.w && o.bias && o.gamma && o.partials_in && o.n_partials_in >= 1, "attn1d + qkv projection: needs the folded "
                    "weights, bias, weight row sums (gamma) and the input's partials");
            REQUIRE(o.CoutPad >= 3 * o.Cin && o.i2 >= 1 && o.i2 <= o.Cin && o.f0 > 0.f,
                    "attn1d + qkv projection: CoutPad=%d < 3 C, or channels i2=%d / eps=%g out of range", o.CoutPad, o.i2, (double)o.f0);
        }
        return ADK_OK;
    case ADK_OP_PACK_INPUT:
        if ((rc = check_geo(o, "pack_input"))) return rc;
        if (o.flags & ADK_FLAG_NO_EMBED) REQUIRE(o.Cin >= 1 && o.x && o.y && !o.gamma, "pack_input: null tensor");
        else REQUIRE(o.Cin >= 1 && o.x && o.w && o.y && o.aux0, "pack_input: null tensor");
        REQUIRE(!o.residual || (o.Cout >= 1 && o.i1 >= 1), "pack_input: cond needs Cc >= 1 and Lc >= 1 (got %d, %d)", o.Cout, o.i1);
        REQUIRE(!o.gamma || o.residual, "pack_input: noise-aug levels without a cond tensor");
        return ADK_OK;
    case ADK_OP_UNPACK_OUTPUT:
        if ((rc = check_geo(o, "unpack_output"))) return rc;
        REQUIRE(o.Cin >= 1 && o.x && o.y, "unpack_output: null tensor");
        REQUIRE(!o.gamma || (o.beta && o.residual), "unpack_output: c_out needs c_skip and the unscaled input");
        return ADK_OK;
    case ADK_OP_LINEAR:
        REQUIRE(o.B >= 1 && o.B <= 65535 && o.Cin >= 1 && o.Cout >= 1 && o.x && o.w && o.y, "linear: bad arguments");
        REQUIRE(o.i0 >= o.Cin && o.i2 >= o.Cout && (!o.residual || o.i1 >= o.Cout), "linear: row strides (%d, %d, %d) "
                "smaller than the rows (%d, %d)", o.i0, o.i1, o.i2, o.Cin, o.Cout);
        return ADK_OK;
    case ADK_OP_FOURIER:
        REQUIRE(o.B >= 1 && o.B <= 65535 && o.Cout >= 2 && o.Cout % 2 == 0 && o.x && o.w && o.y, "fourier: bad arguments");
        REQUIRE(o.i2 >= o.Cout, "fourier: row stride %d < %d features", o.i2, o.Cout);
        return ADK_OK;
    case ADK_OP_PQMF_ANALYSIS:
    case ADK_OP_PQMF_SYNTHESIS: {
        const int M = o.i0;
        REQUIRE(M >= 2 && M <= 64 && (M & (M - 1)) == 0, "pqmf: M=%d must be a power of two in [2,64]", M);
        REQUIRE(o.B >= 1 && o.Cin >= 1 && o.L >= 1 && o.x && o.w && o.y, "pqmf: bad arguments");
        REQUIRE(o.taps >= 1 && (o.taps & 1), "pqmf: filter length %d must be odd", o.taps);
        if (o.kind == ADK_OP_PQMF_ANALYSIS) {
            REQUIRE(o.L % M == 0, "pqmf_analysis: T=%d not a multiple of M=%d", o.L, M);
            REQUIRE(o.taps <= 2049, "pqmf_analysis: K=%d too long", o.taps);
        } else REQUIRE(o.taps <= 129, "pqmf_synthesis: Kt=%d too long", o.taps);
        return ADK_OK;
    }
    case ADK_OP_DDIM_UPDATE:
        REQUIRE(o.x && o.y && o.aux0 && o.aux1, "ddim_update: null tensor");
        REQUIRE(!(o.flags & ADK_FLAG_NOISE) || o.residual, "ddim_update: noise flag without a noise tensor");
        REQUIRE(o.i0 >= 0 && o.i1 >= 0 && (o.i0 > 0 || o.i1 > 0), "ddim_update: n=0");
        return ADK_OK;
    case ADK_OP_STEP_ADVANCE:
        REQUIRE(o.y, "step_advance: null counter");
        return ADK_OK;
    case ADK_OP_SPACE_TO_DEPTH:
    case ADK_OP_DEPTH_TO_SPACE: {
        if ((rc = check_geo(o, "phase_shuffle"))) return rc;
        const int cb = cb_of(o.dtype);
        REQUIRE(o.x && o.y && o.x != o.y, "phase_shuffle: null or aliased tensors");
        REQUIRE(o.i0 >= 1 && o.i0 <= 64, "phase_shuffle: factor %d out of range [1,64]", o.i0);
        REQUIRE(o.Cin >= cb && o.Cin % cb == 0, "phase_shuffle: C=%d must be a multiple of %d", o.Cin, cb);
        REQUIRE(o.P <= 64, "phase_shuffle: halo %d too large", o.P);
        if (o.kind == ADK_OP_SPACE_TO_DEPTH) REQUIRE(o.L % o.i0 == 0, "space_to_depth: L=%d not a multiple of %d", o.L, o.i0);
        REQUIRE((long long)o.B * ((long long)o.L * o.i0 + 2 * o.P) < (1LL << 25), "phase_shuffle: too many rows");
        return ADK_OK;
    }
    case ADK_OP_AUDIO_TO_INT16:
        REQUIRE(o.B >= 1 && o.Cin >= 1 && o.L >= 1 && o.x && o.y, "audio_to_int16: bad arguments");
        REQUIRE((long long)o.B * o.Cin <= 65535, "audio_to_int16: B*C=%lld exceeds 65535", (long long)o.B * o.Cin);
        return ADK_OK;
    default:
        return fail(ADK_EINVAL, "unknown op kind %d", o.kind);
    }
}

static int dispatch(const adk_op& o, hipStream_t st)
{
    switch (o.kind) {
    case ADK_OP_CONV: return launch_conv(o, st);
    case ADK_OP_GN_APPLY: return launch_gn(o, st, true);
    case ADK_OP_GN_STATS: return launch_gn(o, st, false);
    case ADK_OP_FIR_DOWN: return launch_fir(o, st, false);
    case ADK_OP_FIR_UP: return launch_fir(o, st, true);
    case ADK_OP_ATTN: return launch_attn(o, st);
    case ADK_OP_PACK_INPUT: return launch_pack_input(o, st);
    case ADK_OP_UNPACK_OUTPUT: return launch_unpack_output(o, st);
    case ADK_OP_PQMF_ANALYSIS: return launch_pqmf(o, st, false);
    case ADK_OP_PQMF_SYNTHESIS: return launch_pqmf(o, st, true);
    case ADK_OP_DDIM_UPDATE: return launch_ddim_update(o, st);
    case ADK_OP_STEP_ADVANCE: return launch_step_advance(o, st);
    case ADK_OP_AUDIO_TO_INT16: return launch_audio_to_int16(o, st);
    case ADK_OP_SPACE_TO_DEPTH: return launch_phase_shuffle(o, st, true);
    case ADK_OP_DEPTH_TO_SPACE: return launch_phase_shuffle(o, st, false);
    case ADK_OP_LINEAR: return launch_linear(o, st);
    case ADK_OP_FOURIER: return launch_fourier(o, st);
    }
    return ADK_EINVAL;
}

extern "C" {

int adk_version(void) { return 100; }

const char* adk_last_error(void) { return g_err; }

int adk_run_plan(const adk_op* ops, int n, void* stream)
{
    if (!ops || n < 0) return fail(ADK_EINVAL, "adk_run_plan: null plan");
    for (int i = 0; i < n; ++i) {            // validate everything first: all-or-nothing
        const int rc = validate(ops[i]);
        if (rc) {
            char tmp[400];
            snprintf(tmp, sizeof tmp, "%s", g_err);
            return fail(rc, "op %d: %s", i, tmp);
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        const int rc = dispatch(ops[i], st);
        if (rc) return fail(rc, "op %d (kind %d): launch failed: %s", i, ops[i].kind, hipGetErrorString(hipGetLastError()));
    }
    return ADK_OK;
}

int adk_validate_plan(const adk_op* ops, int n)
{
    if (!ops || n < 0) return fail(ADK_EINVAL, "adk_validate_plan: null plan");
    for (int i = 0; i < n; ++i) {
        const int rc = validate(ops[i]);
        if (rc) {
            char tmp[400];
            snprintf(tmp, sizeof tmp, "%s", g_err);
            return fail(rc, "op %d: %s", i, tmp);
        }
    }
    return ADK_OK;
}

int adk_partials_count(const adk_op* op)
{
    if (!op) return 0;
    if (op->kind == ADK_OP_CONV) return conv_partials_count(*op);
    if (op->kind == ADK_OP_GN_APPLY || op->kind == ADK_OP_GN_STATS) return gn_partials_count(op->dtype, op->Cin, op->L);
    return 0;
}

int adk_conv_cotile_used(const adk_op* op) { return (op && op->kind == ADK_OP_CONV) ? conv_cotile_used(*op) : 0; }

int adk_conv_postile(const adk_op* op) { return (op && op->kind == ADK_OP_CONV) ? conv_postile(*op) : 0; }

int adk_conv_cotile(int dtype, int Cout, int L) { (void)dtype; (void)L; return conv_cotile(Cout); }

int adk_sizeof_op(void) { return (int)sizeof(adk_op); }
int adk_conv_gn_fusable(const adk_op* op) { return (op && op->kind == ADK_OP_CONV) ? conv_gn_fusable_here(*op) : 0; }

static adk_op zero_op(int kind, int dtype)
{
    adk_op o;
    memset(&o, 0, sizeof o);
    o.kind = kind; o.dtype = dtype;
    return o;
}

int adk_conv1d(int dtype, uint32_t flags, int B, int L, int P, int Cin, int Cout, int CoutPad, int taps,
               const void* x, const void* w, const float* bias, const void* residual, void* y,
               float* partials_out, void* stream)
{
    adk_op o = zero_op(ADK_OP_CONV, dtype);
    o.flags = flags; o.B = B; o.L = L; o.P = P; o.Cin = Cin; o.Cout = Cout; o.CoutPad = CoutPad; o.taps = taps;
    o.x = x; o.w = w; o.bias = bias; o.residual = residual; o.y = y; o.partials_out = partials_out;
    return adk_run_plan(&o, 1, stream);
}

int adk_gn_apply(int dtype, uint32_t flags, int B, int L, int P, int C, float eps,
                 const void* x, const float* partials_in, int n_partials_in,
                 const float* gamma, const float* beta, const void* residual, void* y,
                 float* partials_out, void* stream)
{
    adk_op o = zero_op(ADK_OP_GN_APPLY, dtype);
    o.flags = flags; o.B = B; o.L = L; o.P = P; o.Cin = C; o.Cout = C; o.f0 = eps;
    o.x = x; o.partials_in = partials_in; o.n_partials_in = n_partials_in; o.gamma = gamma; o.beta = beta;
    o.residual = residual; o.y = y; o.partials_out = partials_out;
    return adk_run_plan(&o, 1, stream);
}

int adk_gn_stats(int dtype, int B, int L, int P, int C, const void* x, float* partials_out, void* stream)
{
    adk_op o = zero_op(ADK_OP_GN_STATS, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.x = x; o.partials_out = partials_out;
    return adk_run_plan(&o, 1, stream);
}

static int fir_entry(int kind, int dtype, int B, int L, int P, int C, int K, const float* taps,
                     const void* x, void* y, void* stream)
{
    adk_op o = zero_op(kind, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.taps = K; o.w = taps; o.x = x; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_fir_down2(int dtype, int B, int L, int P, int C, int K, const float* taps, const void* x, void* y, void* stream)
{
    return fir_entry(ADK_OP_FIR_DOWN, dtype, B, L, P, C, K, taps, x, y, stream);
}

int adk_fir_up2(int dtype, int B, int L, int P, int C, int K, const float* taps, const void* x, void* y, void* stream)
{
    return fir_entry(ADK_OP_FIR_UP, dtype, B, L, P, C, K, taps, x, y, stream);
}

int adk_attn1d(int dtype, int B, int L, int P, int C, const void* qkv, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_ATTN, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.x = qkv; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_attn1d_hd(int dtype, int B, int L, int P, int C, int head_dim, const void* qkv, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_ATTN, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.i0 = head_dim; o.x = qkv; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_adagn_apply(int dtype, uint32_t flags, int B, int L, int P, int C, int groups, float eps,
                    const void* x, const float* partials_in, int n_partials_in,
                    const float* weight, const float* bias, int row_stride, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_GN_APPLY, dtype);
    o.flags = flags | ADK_FLAG_ADA; o.B = B; o.L = L; o.P = P; o.Cin = C; o.Cout = C; o.f0 = eps;
    o.i0 = row_stride; o.i1 = groups;
    o.x = x; o.partials_in = partials_in; o.n_partials_in = n_partials_in; o.gamma = weight; o.beta = bias; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_linear(uint32_t flags, int B, int Cin, int Cout, const float* x, int x_stride, const float* w, const float* bias,
               const float* residual, int res_stride, float* y, int y_stride, void* stream)
{
    adk_op o = zero_op(ADK_OP_LINEAR, ADK_F32);
    o.flags = flags; o.B = B; o.Cin = Cin; o.Cout = Cout; o.x = x; o.i0 = x_stride; o.w = w; o.bias = bias;
    o.residual = residual; o.i1 = res_stride; o.y = y; o.i2 = y_stride;
    return adk_run_plan(&o, 1, stream);
}

int adk_fourier_features(uint32_t flags, int B, int F, const float* sigma, const float* w, float* y, int y_stride, void* stream)
{
    adk_op o = zero_op(ADK_OP_FOURIER, ADK_F32);
    o.flags = flags; o.B = B; o.Cout = F; o.x = sigma; o.w = w; o.y = y; o.i2 = y_stride;
    return adk_run_plan(&o, 1, stream);
}

int adk_pack_input_plain(int dtype, int B, int L, int P, int Cx, const float* x, const float* x_scale,
                         const float* cond, int Cc, int Lc, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_PACK_INPUT, dtype);
    o.flags = ADK_FLAG_NO_EMBED; o.B = B; o.L = L; o.P = P; o.Cin = Cx; o.x = x; o.beta = x_scale; o.y = y;
    o.residual = cond; o.Cout = Cc; o.i1 = Lc;
    return adk_run_plan(&o, 1, stream);
}

int adk_unpack_output_scaled(int B, int L, int P, int C, const float* x, const float* c_out, const float* c_skip,
                             const float* x_in, float* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_UNPACK_OUTPUT, ADK_F32);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.x = x; o.gamma = c_out; o.beta = c_skip; o.residual = x_in; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_pack_input(int dtype, int B, int L, int P, int Cx, const float* x, const float* t,
                   const int32_t* step, int t_stride, const float* fw, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_PACK_INPUT, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = Cx; o.x = x; o.aux0 = t; o.bias = reinterpret_cast<const float*>(step);
    o.i0 = t_stride; o.w = fw; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_pack_input_cond(int dtype, int B, int L, int P, int Cx, const float* x, const float* t,
                        const int32_t* step, int t_stride, const float* fw,
                        const float* cond, int Cc, int Lc, const float* aug_level, const float* aug_noise,
                        void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_PACK_INPUT, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = Cx; o.x = x; o.aux0 = t; o.bias = reinterpret_cast<const float*>(step);
    o.i0 = t_stride; o.w = fw; o.y = y;
    o.residual = cond; o.Cout = Cc; o.i1 = Lc; o.gamma = aug_level; o.aux1 = const_cast<float*>(aug_noise);
    return adk_run_plan(&o, 1, stream);
}

int adk_unpack_output(int B, int L, int P, int C, const float* x, float* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_UNPACK_OUTPUT, ADK_F32);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.x = x; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_pqmf_analysis(int B, int Ch, int T, int M, int K, const float* x, const float* fwT, float* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_PQMF_ANALYSIS, ADK_F32);
    o.B = B; o.Cin = Ch; o.L = T; o.i0 = M; o.taps = K; o.x = x; o.w = fwT; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_pqmf_synthesis(int B, int Ch, int N, int M, int Kt, const float* x, const float* iwT, float* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_PQMF_SYNTHESIS, ADK_F32);
    o.B = B; o.Cin = Ch; o.L = N; o.i0 = M; o.taps = Kt; o.x = x; o.w = iwT; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_ddim_update(uint32_t flags, int64_t n, float* x, const float* v, const float* noise, float* pred,
                    const float* coef, const int32_t* step, void* stream)
{
    adk_op o = zero_op(ADK_OP_DDIM_UPDATE, ADK_F32);
    o.flags = flags; o.i0 = (int32_t)(n & 0x7fffffff); o.i1 = (int32_t)(n >> 31);
    o.y = x; o.x = v; o.residual = noise; o.aux1 = pred; o.aux0 = coef; o.bias = reinterpret_cast<const float*>(step);
    return adk_run_plan(&o, 1, stream);
}

int adk_space_to_depth(int dtype, int B, int L, int P, int C, int factor, const void* x, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_SPACE_TO_DEPTH, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.i0 = factor; o.x = x; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_depth_to_space(int dtype, int B, int L, int P, int C, int factor, const void* x, void* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_DEPTH_TO_SPACE, dtype);
    o.B = B; o.L = L; o.P = P; o.Cin = C; o.i0 = factor; o.x = x; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_audio_to_int16(int B, int C, int T, const float* x, int16_t* y, void* stream)
{
    adk_op o = zero_op(ADK_OP_AUDIO_TO_INT16, ADK_F32);
    o.B = B; o.Cin = C; o.L = T; o.x = x; o.y = y;
    return adk_run_plan(&o, 1, stream);
}

int adk_step_advance(int32_t* step, void* stream)
{
    adk_op o = zero_op(ADK_OP_STEP_ADVANCE, ADK_F32);
    o.y = step;
    return adk_run_plan(&o, 1, stream);
}

// Address of this device's copy of `scratch` (looked up once per device, with that device current).
static int scratch_for_current_device(hipStream_t st, float** out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ADK_MAX_DEVICES)
        return fail(ADK_ELAUNCH, "group_norm_noalloc: no current HIP device (or its index is >= %d)", ADK_MAX_DEVICES);
    if (st) {                                    // the scratch copy used must be the one of the device that runs the kernels
        hipDevice_t sdev;
        if (hipStreamGetDevice(st, &sdev) == hipSuccess && (int)sdev != dev)
            return fail(ADK_EINVAL, "group_norm_noalloc: the stream belongs to device %d but device %d is current", (int)sdev, dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(ADK_ELAUNCH, "group_norm_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                        hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return ADK_OK;
}

int adk_group_norm_noalloc(int dtype, uint32_t flags, int B, int L, int P, int C, float eps,
                           const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                           void* stream)
{
    adk_op s = zero_op(ADK_OP_GN_STATS, dtype);
    s.B = B; s.L = L; s.P = P; s.Cin = C; s.x = x;
    s.partials_out = reinterpret_cast<float*>(1);        // placeholder for validation; patched below
    int rc = validate(s);
    if (rc) return rc;
    const int n = gn_partials_count(dtype, C, L);
    if ((long long)B * n > ADK_NOALLOC_MAX_PAIRS)
        return fail(ADK_EINVAL, "group_norm_noalloc: B * partials = %d * %d pairs exceed the built-in scratch of %d pairs "
                    "(use adk_gn_stats + adk_gn_apply with a caller-owned buffer)", B, n, ADK_NOALLOC_MAX_PAIRS);
    adk_op o = zero_op(ADK_OP_GN_APPLY, dtype);
    o.flags = flags & ADK_FLAG_GELU; o.B = B; o.L = L; o.P = P; o.Cin = C; o.Cout = C; o.f0 = eps;
    o.x = x; o.n_partials_in = n; o.gamma = gamma; o.beta = beta; o.residual = residual; o.y = y;
    o.partials_in = reinterpret_cast<const float*>(1);
    if ((rc = validate(o))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ptr = nullptr;
    if ((rc = scratch_for_current_device(st, &ptr))) return rc;
    s.partials_out = ptr;                        // the device address, never the array's name
    o.partials_in = ptr;
    // nothing has to start at zero: the statistics kernel overwrites all B * n pairs the apply kernel reads
    if ((rc = dispatch(s, st)) || (rc = dispatch(o, st)))
        return fail(rc, "group_norm_noalloc: launch failed: %s", hipGetErrorString(hipGetLastError()));
    return ADK_OK;
}

}  // extern "C"
