// C-ABI layer (include/aefft.h): context, workspaces, op-level entry points and the resident
// batched network.  Host-side orchestration only -- all arithmetic lives in the *_kernels.hip files.
#include "../../include/aefft.h"
#include "internal.h"

#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace aefft;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
enum { WS_MID = 0, WS_REAL = 1, WS_S = 2, WS_ES = 3, WS_E = 4, WS_DC = 5, WS_DF = 6, WS_SMALL = 7, WS_DEN = 8, WS_TMP = 9, WS_PART = 10, WS_MID2 = 11, WS_MID3 = 12, WS_COUNT = 13 };

// fine-grained kernel ids for profiling; the public classes (aefft.h) aggregate them
enum {
    KID_R2C_ROWS = 0, KID_R2C_COLS, KID_C2R_COLS, KID_C2R_ROWS, KID_CONTRACT, KID_RESIZE, KID_DIFFMSE, KID_BIASGRAD,
    KID_PAD, KID_SHRINK, KID_UPDATE, KID_GDIFF, KID_SPATIAL, KID_KSPEC, KID_KGRAD, KID_WGRAD, KID_OPFORM, KID_CHAIN, KID_SGRAD, KID_OPMSE, KID_COUNT
};

struct ProfEvent { hipEvent_t a, b; int kid; double bytes; };

struct aefft_ctx {
    int device = 0;
    hipStream_t stream = nullptr;    // the caller-visible stream: every public call is ordered on it
    hipStream_t cur = nullptr;       // stream the helpers enqueue on (== stream except inside a forked section)
    bool own_stream = false;
    bool in_u8 = false;              // the frames handed to the running call are 8-bit pixels (aefft_net_step_grad_u8 / aefft_net_forward_u8: do_r2c converts on load)
    int biasColP1 = 0;               // operator form: conv_k biases go to the affine column of the basis frames only (Contract::biasColP1)
    bool recon_join = false;         // a deferred reconstruction (pipelined mode) still has to be joined from aux[0] (ev_join[0])
    static const int NAUX = 2;
    hipStream_t aux[NAUX] = {};      // side streams: 0 = reconstruction inverse FFT, 1 = input prefetch (created with the first net)
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {};
    int side_cus = 0;                // aefft_ctx_partition: CUs of the side streams (0: no partition)
    std::string err;
    const float2* tw = nullptr;      // device twiddle table
    void* ws[WS_COUNT] = {};
    size_t ws_bytes[WS_COUNT] = {};
    bool prof = false;
    std::vector<ProfEvent> pool;     // pre-created events
    size_t used = 0;
    long launches[KID_COUNT] = {};
    double ms[KID_COUNT] = {};
    double bytes[KID_COUNT] = {};
};

static int fail(aefft_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        ctx->err = buf;
    }
    return code;
}
#define HIPCHK(ctx, call)                                                     \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return fail(ctx, AEFFT_EHIP, #call, e_);        \
    } while (0)
#define RET_IF(x)                    \
    do {                             \
        int r_ = (x);                \
        if (r_ != AEFFT_OK) return r_; \
    } while (0)

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static int ws_get(aefft_ctx* ctx, int slot, size_t bytes, void** out)
{
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr; ctx->ws_bytes[slot] = 0;
        }
        size_t want = (bytes + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&ctx->ws[slot], want);
        if (e != hipSuccess) return fail(ctx, AEFFT_ENOMEM, "hipMalloc(workspace)", e);
        if (flag(AEFFT_F_POISON)) { HIPCHK(ctx, hipMemset(ctx->ws[slot], 0xFF, want)); HIPCHK(ctx, hipDeviceSynchronize()); }   // NaN-fill: uninitialised reads show up in the tests
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return AEFFT_OK;
}

// profiling brackets --------------------------------------------------------------------------
struct Bracket {
    aefft_ctx* ctx; int idx = -1;
    Bracket(aefft_ctx* c, int kid, double bytes) : ctx(c)
    {
        if (!c->prof) return;
        if (c->used >= c->pool.size()) return;   // pool exhausted: stop recording (read() reports what it has)
        idx = (int)c->used++;
        c->pool[idx].kid = kid; c->pool[idx].bytes = bytes;
        (void)hipEventRecord(c->pool[idx].a, c->cur);
    }
    ~Bracket() { if (idx >= 0) (void)hipEventRecord(ctx->pool[idx].b, ctx->cur); }
};

namespace aefft { unsigned dev_flags = 0; }
static const struct { const char* name; unsigned bit; } flag_names[] = {
    {"NOLAZY", AEFFT_F_NOLAZY}, {"NOCOMPACT", AEFFT_F_NOCOMPACT}, {"NOQPATH", AEFFT_F_NOQPATH}, {"NOFUSEMSE", AEFFT_F_NOFUSEMSE},
    {"NOGROUP", AEFFT_F_NOGROUP}, {"NOMFMA", AEFFT_F_NOMFMA}, {"NOGFWD", AEFFT_F_NOGFWD}, {"NOOVERLAP", AEFFT_F_NOOVERLAP},
    {"NOFUSECROP", AEFFT_F_NOFUSECROP}, {"GTAPS", AEFFT_F_GTAPS}, {"NOPREFETCH", AEFFT_F_NOPREFETCH}, {"NODEFER", AEFFT_F_NODEFER},
    {"NOTILEDSPATIAL", AEFFT_F_NOTILEDSPATIAL}, {"NOFAST", AEFFT_F_NOFAST}, {"NOSPLITK", AEFFT_F_NOSPLITK}, {"POISON", AEFFT_F_POISON},
    {"NOOPFORM", AEFFT_F_NOOPFORM}, {"NOCHAIN", AEFFT_F_NOCHAIN}, {"NOFUSEUPD", AEFFT_F_NOFUSEUPD}, {"NOAHEAD", AEFFT_F_NOAHEAD}, {"NORCORR", AEFFT_F_NORCORR}, {"NOLAZYMSE", AEFFT_F_NOLAZYMSE},
    {"SMALLOVERLAP", AEFFT_F_SMALLOVERLAP}, {"CHAINMSE", AEFFT_F_CHAINMSE}};
// The switches named by AEFFT_FLAGS stay on for the life of the process: aefft_ctx_set_flags ORs its argument onto them (a test fixture
// that restores "no flags" does not clear an AEFFT_FLAGS=POISON run).  A name the library does not know is an error, not a silent
// default run: the first aefft_ctx_create fails with AEFFT_EINVAL and says which.
static unsigned env_flags = 0;
static std::string env_flags_error;
static void flags_from_env_once()
{
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("AEFFT_FLAGS");           // the ONLY environment lookup of the library
    if (!e) return;
    std::string s(e);
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string w = s.substr(i, j - i);
        bool known = w.empty();
        for (const auto& f : flag_names) if (w == f.name) { env_flags |= f.bit; known = true; }
        if (!known) env_flags_error += (env_flags_error.empty() ? "" : ",") + w;
        i = j + 1;
    }
    dev_flags = env_flags;
}
extern "C" int aefft_ctx_set_flags(aefft_ctx* ctx, unsigned flags) { if (!ctx) return AEFFT_EINVAL; dev_flags = env_flags | flags; return AEFFT_OK; }
extern "C" unsigned aefft_ctx_get_flags(const aefft_ctx*) { return dev_flags; }

extern "C" const char* aefft_version(void) { return "aefft 0.2 (gfx950)"; }

extern "C" int aefft_ctx_create(aefft_ctx** out, int device, void* hip_stream, int create_stream)
{
    if (!out) return AEFFT_EINVAL;
    *out = nullptr;
    flags_from_env_once();
    if (!env_flags_error.empty()) { fprintf(stderr, "aefft: unknown name(s) in AEFFT_FLAGS: %s\n", env_flags_error.c_str()); return AEFFT_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return AEFFT_EHIP;
    aefft_ctx* ctx = new aefft_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return AEFFT_EHIP; }
    if (!create_stream) ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
    else {
        // the library's own stream carries the latency-bound chain of the step: highest priority, so that the bandwidth-bound
        // side-stream work (reconstruction, input prefetch) takes the slots it leaves free instead of crowding it out
        int pr_least = 0, pr_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
        if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, pr_greatest) != hipSuccess) { delete ctx; return AEFFT_EHIP; }
        ctx->own_stream = true;
    }
    if (upload_twiddles(ctx->stream) != hipSuccess) { if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream); delete ctx; return AEFFT_EHIP; }
    ctx->cur = ctx->stream;
    ctx->tw = twiddle_table();
    *out = ctx;
    return AEFFT_OK;
}

extern "C" void aefft_ctx_destroy(aefft_ctx* ctx)
{
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < WS_COUNT; ++i) if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    for (auto& e : ctx->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int i = 0; i < aefft_ctx::NAUX; ++i) { if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); } if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// CU mask of one side of the partition: the side streams take mask bits [0, side_cus), the context stream the rest.  (Mask bits are dealt
// to the XCDs round-robin by the driver -- tools/cumask_probe.hip -- so a run of consecutive bits is the same share of every XCD.)
static hipError_t masked_stream(hipStream_t* st, int ncu, int lo, int hi)
{
    std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u);
    for (int i = lo; i < hi; ++i) m[(size_t)i / 32] |= 1u << (i % 32);
    return hipExtStreamCreateWithCUMask(st, (uint32_t)m.size(), m.data());
}
static int device_cus(int device)
{
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return n;
}

// side streams (0: reconstruction inverse FFT, 1: input prefetch) and their events; all-or-nothing
static int ensure_aux(aefft_ctx* ctx)
{
    if (ctx->aux[0]) return AEFFT_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < aefft_ctx::NAUX && e == hipSuccess; ++i) {
        int pr_least = 0, pr_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
        if (ctx->side_cus > 0) e = masked_stream(&ctx->aux[i], device_cus(ctx->device), 0, ctx->side_cus);
        else
        e = hipStreamCreateWithPriority(&ctx->aux[i], hipStreamNonBlocking, pr_least);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) return AEFFT_OK;
    for (int i = 0; i < aefft_ctx::NAUX; ++i) {
        if (ctx->aux[i]) (void)hipStreamDestroy(ctx->aux[i]);
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
        ctx->aux[i] = nullptr; ctx->ev_join[i] = nullptr;
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    ctx->ev_fork = nullptr;
    return fail(ctx, AEFFT_EHIP, "side streams", e);
}

extern "C" int aefft_ctx_partition(aefft_ctx* ctx, int side_cus)
{
    if (!ctx) return AEFFT_EINVAL;
    if (!ctx->own_stream || ctx->aux[0]) return fail(ctx, AEFFT_ESTATE, "aefft_ctx_partition: needs a context that owns its stream, before its first net");
    const int ncu = device_cus(ctx->device);
    if (side_cus < 0 || (side_cus > 0 && (ncu < 16 || side_cus < 8 || side_cus > ncu - 8))) return fail(ctx, AEFFT_EINVAL, "aefft_ctx_partition: side_cus out of range");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    hipStream_t st = nullptr;
    if (side_cus > 0) HIPCHK(ctx, masked_stream(&st, ncu, side_cus, ncu));
    else {
        int pr_least = 0, pr_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
        HIPCHK(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr_greatest));
    }
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = ctx->cur = st;
    ctx->side_cus = side_cus;
    return AEFFT_OK;
}

extern "C" const char* aefft_last_error(const aefft_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
static int join_recon(aefft_ctx* ctx)
{
    if (!ctx->recon_join) return AEFFT_OK;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[0], 0));
    ctx->recon_join = false;
    return AEFFT_OK;
}
extern "C" int aefft_sync(aefft_ctx* ctx) { if (!ctx) return AEFFT_EINVAL; RET_IF(join_recon(ctx)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return AEFFT_OK; }
extern "C" void* aefft_stream(aefft_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int aefft_prof_enable(aefft_ctx* ctx, int enable)
{
    if (!ctx) return AEFFT_EINVAL;
    if (enable && ctx->pool.empty()) {
        ctx->pool.resize(8192);
        for (auto& e : ctx->pool) { HIPCHK(ctx, hipEventCreate(&e.a)); HIPCHK(ctx, hipEventCreate(&e.b)); }
    }
    ctx->prof = enable != 0;
    return AEFFT_OK;
}

static int prof_collect(aefft_ctx* ctx)
{
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < aefft_ctx::NAUX; ++i) if (ctx->aux[i]) HIPCHK(ctx, hipStreamSynchronize(ctx->aux[i]));
    for (size_t i = 0; i < ctx->used; ++i) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->pool[i].a, ctx->pool[i].b));
        const int k = ctx->pool[i].kid;
        ctx->launches[k]++; ctx->ms[k] += ms; ctx->bytes[k] += ctx->pool[i].bytes;
    }
    ctx->used = 0;
    return AEFFT_OK;
}

extern "C" int aefft_prof_reset(aefft_ctx* ctx)
{
    if (!ctx) return AEFFT_EINVAL;
    RET_IF(prof_collect(ctx));
    memset(ctx->launches, 0, sizeof ctx->launches); memset(ctx->ms, 0, sizeof ctx->ms); memset(ctx->bytes, 0, sizeof ctx->bytes);
    return AEFFT_OK;
}

static const char* kid_names[KID_COUNT] = {"r2c_rows", "r2c_cols", "c2r_cols", "c2r_rows", "contract", "resize", "diff_mse",
                                           "bias_grad", "pad", "shrink", "update", "gradient_diff", "spatial", "kspec", "kgrad", "weight_taps", "moment", "chain", "sgrad", "opmse"};

extern "C" int aefft_prof_read(aefft_ctx* ctx, int kid, long* launches, double* total_ms, double* algo_bytes)
{
    if (!ctx || kid < 0 || kid >= KID_COUNT) return AEFFT_EINVAL;
    RET_IF(prof_collect(ctx));
    if (launches) *launches = ctx->launches[kid];
    if (total_ms) *total_ms = ctx->ms[kid];
    if (algo_bytes) *algo_bytes = ctx->bytes[kid];
    return AEFFT_OK;
}
extern "C" const char* aefft_prof_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kid_names[kid] : nullptr; }
extern "C" int aefft_prof_count(void) { return KID_COUNT; }

// ------------------------------------------------------------------------------------------
// internal op helpers (all enqueue on ctx->cur)
// ------------------------------------------------------------------------------------------
static long bins(int Nx, int Ny) { return (long)Nx * (Ny / 2 + 1); }

static int chk_size(aefft_ctx* ctx, int Nx, int Ny)
{
    if (!fft_size_supported(Nx) || !fft_size_supported(Ny)) return fail(ctx, AEFFT_EINVAL, "Nx, Ny must be powers of two in 8..2048");
    return AEFFT_OK;
}

static int chk_size_any(aefft_ctx* ctx, int Nx, int Ny)
{
    if ((fft_size_supported(Nx) && fft_size_supported(Ny)) || (fft_size_supported_any(Nx) && fft_size_supported_any(Ny) && Nx <= 1024 && Ny <= 1024)) return AEFFT_OK;
    return fail(ctx, AEFFT_EINVAL, "Nx, Ny must be powers of two in 8..2048, or even sizes in 8..1024");
}
static bool pow2_sizes(int Nx, int Ny) { return fft_size_supported(Nx) && fft_size_supported(Ny); }
static int do_resize(aefft_ctx* ctx, const float2* in, float2* out, long planes, int Nx, int Ny, int Nxs, int Nys);

// sizes that are not powers of two (fft_backproplib.cu:773-779: cufftPlanMany takes any): Bluestein rows + transposes (fft_kernels.hip);
// the spectral crop / zero-pad as a separate resize
static int do_r2c_any(aefft_ctx* ctx, const float* x, float2* X, long planes, int Nx, int Ny, int Nxs, int Nys)
{
    const size_t el = fft_any_ws_elems(planes, Nx, Ny);
    void *w1, *w2, *w3 = nullptr;
    RET_IF(ws_get(ctx, WS_MID, sizeof(float2) * el, &w1));
    RET_IF(ws_get(ctx, WS_MID2, sizeof(float2) * el, &w2));
    const bool crop = Nxs != Nx || Nys != Ny;
    if (crop) RET_IF(ws_get(ctx, WS_MID3, sizeof(float2) * el, &w3));
    {
        Bracket br(ctx, KID_R2C_ROWS, (double)planes * ((double)Nx * Ny * 4.0 + (double)bins(Nx, Ny) * 8.0));
        hipError_t e = launch_r2c_any(x, crop ? (float2*)w3 : X, (float2*)w1, (float2*)w2, planes, Nx, Ny, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "r2c (any size)", e);
    }
    return crop ? do_resize(ctx, (const float2*)w3, X, planes, Nx, Ny, Nxs, Nys) : AEFFT_OK;
}
static int do_c2r_any(aefft_ctx* ctx, const float2* X, float* x, long planes, int Nxi, int Nyi, int Nx, int Ny, float scale)
{
    const size_t el = fft_any_ws_elems(planes, Nx, Ny);
    void *w1, *w2, *w3 = nullptr;
    RET_IF(ws_get(ctx, WS_MID, sizeof(float2) * el, &w1));
    RET_IF(ws_get(ctx, WS_MID2, sizeof(float2) * el, &w2));
    const bool pad = Nxi != Nx || Nyi != Ny;
    if (pad) {
        RET_IF(ws_get(ctx, WS_MID3, sizeof(float2) * el, &w3));
        RET_IF(do_resize(ctx, X, (float2*)w3, planes, Nxi, Nyi, Nx, Ny));
    }
    Bracket br(ctx, KID_C2R_ROWS, (double)planes * ((double)Nx * Ny * 4.0 + (double)bins(Nx, Ny) * 8.0));
    hipError_t e = launch_c2r_any(pad ? (const float2*)w3 : X, x, (float2*)w1, (float2*)w2, planes, Nx, Ny, scale, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "c2r (any size)", e);
    return AEFFT_OK;
}

// R2C (+ fused crop to Nxs x Nys).  The two kernels are bracketed separately for profiling.
// (ctx->in_u8, set by aefft_net_step_grad_u8 for the duration of its call: x holds 8-bit pixels)
static int do_r2c(aefft_ctx* ctx, const float* x, float2* X, long planes, int Nx, int Ny, int Nxs, int Nys, int ws_id = WS_MID, hipEvent_t done = nullptr)
{
    const bool u8 = ctx->in_u8;
    if (!pow2_sizes(Nx, Ny) || !pow2_sizes(Nxs, Nys)) {
        if (u8) return fail(ctx, AEFFT_EINVAL, "r2c: 8-bit frames need power-of-two sizes");
        RET_IF(chk_size_any(ctx, Nx, Ny));
        if (!aligned16(x) || !aligned16(X)) return fail(ctx, AEFFT_EINVAL, "r2c: pointers must be 16-byte aligned");
        return do_r2c_any(ctx, x, X, planes, Nx, Ny, Nxs, Nys);
    }
    RET_IF(chk_size(ctx, Nx, Ny));
    if (!aligned16(x) || !aligned16(X)) return fail(ctx, AEFFT_EINVAL, "r2c: pointers must be 16-byte aligned");
    void* mid;
    RET_IF(ws_get(ctx, ws_id, sizeof(float2) * fft_mid_elems(planes, Nx, Nys / 2), &mid));
    // launch_r2c issues rows then cols; bracket as two launches by splitting the byte accounting:
    // rows: read planes*Nx*Ny*4, write mid; cols: read mid, write out.
    const double b_in = (double)planes * Nx * Ny * (u8 ? 1 : 4), b_mid = (double)planes * Nx * (Nys / 2) * 8, b_out = (double)planes * bins(Nxs, Nys) * 8;
    hipError_t e;
    {
        // The row and column kernels are launched inside launch_r2c; to time them separately we call it in two halves.
        Bracket br(ctx, KID_R2C_ROWS, b_in + b_mid);
        e = launch_r2c(x, nullptr, (float2*)mid, planes, Nx, Ny, Nxs, Nys, ctx->cur, nullptr, u8);
    }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "r2c rows", e);
    {
        Bracket br(ctx, KID_R2C_COLS, b_mid + b_out);
        e = launch_r2c(nullptr, X, (float2*)mid, planes, Nx, Ny, Nxs, Nys, ctx->cur, done);
    }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "r2c cols", e);
    return AEFFT_OK;
}

static int do_c2r(aefft_ctx* ctx, const float2* X, float* x, long planes, int Nxi, int Nyi, int Nx, int Ny, float scale, int ws_id = WS_MID,
                  const OpIn* opin = nullptr)
{
    if (!opin && (!pow2_sizes(Nx, Ny) || !pow2_sizes(Nxi, Nyi))) {
        RET_IF(chk_size_any(ctx, Nx, Ny));
        if (!aligned16(x) || !aligned16(X)) return fail(ctx, AEFFT_EINVAL, "c2r: pointers must be 16-byte aligned");
        return do_c2r_any(ctx, X, x, planes, Nxi, Nyi, Nx, Ny, scale);
    }
    RET_IF(chk_size(ctx, Nx, Ny));
    if (!aligned16(x) || (!opin && !aligned16(X))) return fail(ctx, AEFFT_EINVAL, "c2r: pointers must be 16-byte aligned");
    void* mid;
    RET_IF(ws_get(ctx, ws_id, sizeof(float2) * fft_mid_elems(planes, Nx, Nyi / 2), &mid));
    const double b_in = (double)planes * bins(Nxi, Nyi) * 8, b_mid = (double)planes * Nx * (Nyi / 2) * 8, b_out = (double)planes * Nx * Ny * 4;
    hipError_t e;
    {
        Bracket br(ctx, KID_C2R_COLS, b_in + b_mid);
        e = launch_c2r(X, nullptr, (float2*)mid, planes, Nxi, Nyi, Nx, Ny, scale, ctx->cur, opin);
    }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "c2r cols", e);
    {
        Bracket br(ctx, KID_C2R_ROWS, b_mid + b_out);
        e = launch_c2r(nullptr, x, (float2*)mid, planes, Nxi, Nyi, Nx, Ny, scale, ctx->cur);
    }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "c2r rows", e);
    return AEFFT_OK;
}

// algorithmic bytes of one contraction = UNIQUE tensors entering + leaving: A (R*K planes) + B (K*C planes) + Out (R*C planes),
// 8 B per bin; an operand that is the same tensor as another one (X X^H; the MSE epilogue's T = B) counts once; A2 counts.
static double contract_bytes(const Contract& q)
{
    const bool self = q.A == q.B && q.a_r == q.b_c && q.a_k == q.b_k && q.R == q.C;
    double planes = (double)q.R * q.K + (self ? 0.0 : (double)q.K * q.C);
    if (q.A2 && q.A2 != q.B) planes += (double)q.R * q.K;
    if (!q.mse.acc) planes += (double)q.R * q.C;
    return planes * q.P * 8.0;
}

static Contract bc(const aefft_ctx* ctx, Contract q) { if (q.bias) q.biasColP1 = ctx->biasColP1; return q; }

static int do_contract(aefft_ctx* ctx, const Contract& q0)
{
    const Contract q = bc(ctx, q0);
    const double bytes = contract_bytes(q);
    Bracket br(ctx, KID_CONTRACT, bytes);
    hipError_t e = launch_contract(q, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "contract", e);
    return AEFFT_OK;
}

static int do_contract2(aefft_ctx* ctx, const Contract& q0, const Contract& q1)
{
    const double bytes = contract_bytes(q0) + contract_bytes(q1);
    Contract2 qq{};
    qq.q[0] = bc(ctx, q0); qq.q[1] = bc(ctx, q1); qq.n = 2;
    Bracket br(ctx, KID_CONTRACT, bytes);
    hipError_t e = launch_contract2(qq, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "contract2", e);
    return AEFFT_OK;
}

// ---- contraction descriptors (shared by the single-problem and the grouped launches) ----
static Contract mk_conv(const float2* X, const float2* W, const float* bias, float2* O, int B, int R, int K, int Nx, int Ny)
{
    const long P = bins(Nx, Ny);
    Contract q{};
    q.A = W; q.a_r = (long)K * P; q.a_k = P;
    q.B = X; q.b_k = P; q.b_c = (long)K * P;
    q.Out = O; q.o_r = P; q.o_c = (long)R * P;
    q.R = R; q.C = B; q.K = K; q.P = P;
    q.preDivB = (float)R;                       // in_t /= dM   (fft_backproplib.cu:176-177)
    q.bias = bias; q.biasScale = (float)Nx * (float)Ny; q.biasAfterFirst = true;
    return q;
}
static float grad_norm(int dM, int dD, int Nx, int Ny)
{
    const float norm = (float)Nx * (float)Ny;                 // fft_backproplib.cu:398
    return norm * 2 * dM * dD * Nx * Ny;                      // :399 (float arithmetic, left to right)
}
static Contract mk_S(const float2* Xin, const float2* T, const float2* O, float2* S, int B, int dD, long P)
{
    Contract q{};
    q.A = O; q.A2 = T; q.a_r = P; q.a_k = (long)dD * P;
    q.B = Xin; q.b_k = (long)dD * P; q.b_c = P; q.conjB = true;
    q.Out = S; q.o_r = (long)dD * P; q.o_c = P;
    q.R = dD; q.C = dD; q.K = B; q.P = P;
    return q;
}
// The same S when O is stored on the support of the up-sampled spectra only (Oc[b][d][s], s on the small grid):
//   S = -sum_b X_b X_b^H  on every bin,   S[map(s)] += sum_b Oc_b[s] X_b[map(s)]^H  on the support.
static Contract mk_XXneg(const float2* X, float2* S, int B, int dD, long P)
{
    Contract q{};
    q.A = X; q.a_r = P; q.a_k = (long)dD * P;
    q.B = X; q.b_k = (long)dD * P; q.b_c = P; q.conjB = true;
    q.Out = S; q.o_r = (long)dD * P; q.o_c = P;
    q.R = dD; q.C = dD; q.K = B; q.P = P;
    q.postDiv = -1.0f;
    return q;
}
static Contract mk_OX(const float2* Oc, const float2* X, float2* S, int B, int dD, long P, long Pc, int Nx, int Ny, int NxC, int NyC)
{
    Contract q{};
    q.A = Oc; q.a_r = Pc; q.a_k = (long)dD * Pc;
    q.B = X; q.b_k = (long)dD * P; q.b_c = P; q.conjB = true;
    q.Out = S; q.o_r = (long)dD * P; q.o_c = P;
    q.R = dD; q.C = dD; q.K = B; q.P = Pc;
    q.gdNx = Nx; q.gdNy = Ny; q.gdNxs = NxC; q.gdNys = NyC; q.gdMask = 2 | 4;
    q.accumulate = true;
    return q;
}
static Contract mk_dc(const float2* F, const float2* S, float2* dc, int B, int dM, int dD, long P, float Norm)
{
    Contract q{};
    q.A = F; q.a_r = P; q.a_k = (long)dM * P; q.conjA = true;
    q.B = S; q.b_k = (long)dD * P; q.b_c = P;
    q.Out = dc; q.o_r = (long)dD * P; q.o_c = P;
    q.R = dM; q.C = dD; q.K = dD; q.P = P;
    q.postDiv = Norm * (float)B;
    return q;
}
static Contract mk_df(const float2* C, const float2* S, float2* df, int B, int dM, int dD, long P, float Norm)
{
    Contract r{};
    r.A = S; r.a_r = (long)dD * P; r.a_k = P;
    r.B = C; r.b_k = P; r.b_c = (long)dD * P; r.conjB = true;
    r.Out = df; r.o_r = (long)dM * P; r.o_c = P;
    r.R = dD; r.C = dM; r.K = dD; r.P = P;
    r.postDiv = Norm * (float)B;
    return r;
}
// Re-forward of one pair for its MSE only (fft_backproplib.cu:1460-1463 when nothing else consumes H and O): the two
// conv_k collapse per bin into G[d'][d] = sum_m F[d'][m] C[m][d] / (dM*dD) (no batch dimension) ...
static Contract mk_G(const float2* F, const float2* C, float2* G, int dM, int dD, long P)
{
    Contract q{};
    q.A = F; q.a_r = (long)dM * P; q.a_k = P;
    q.B = C; q.b_k = (long)dD * P; q.b_c = P;
    q.Out = G; q.o_r = (long)dD * P; q.o_c = P;
    q.R = dD; q.C = dD; q.K = dM; q.P = P;
    q.postDiv = (float)dM * (float)dD;
    return q;
}
// ... and O_b = G X_b (+ the bias terms at DC) is compared with X_b inside the contraction's epilogue: H and O never exist.
static Contract mk_gmse(const float2* G, const float2* X, const float2* F, const float* b, const float* p, float* mse_slot,
                        int B, int dM, int dD, int Nx, int Ny)
{
    const long P = bins(Nx, Ny);
    Contract q{};
    q.A = G; q.a_r = (long)dD * P; q.a_k = P;
    q.B = X; q.b_k = P; q.b_c = (long)dD * P;
    q.R = dD; q.C = B; q.K = dD; q.P = P;
    q.mse.acc = mse_slot; q.mse.F = F; q.mse.b = b; q.mse.p = p; q.mse.dM = dM; q.mse.Nyr = Ny / 2 + 1;
    q.mse.nfull = (float)dD * Nx * Ny;
    q.mse.scale = 1.0f / ((float)(2 * dM) * (float)Nx * (float)Ny * (float)B);        // as do_diff_mse
    q.mse.norm = (float)Nx * (float)Ny;
    return q;
}
static int do_contract(aefft_ctx* ctx, const Contract& q);
// n independent contractions of class cls (see ContractN) in one launch; falls back to one launch each
static int do_contract_group(aefft_ctx* ctx, const Contract* qs, int n, int nA, int cls)
{
    if (n <= 8 && n > 1 && !flag(AEFFT_F_NOGROUP)) {
        ContractN g{};
        double bytes = 0;
        for (int i = 0; i < n; ++i) { g.q[i] = bc(ctx, qs[i]); bytes += contract_bytes(qs[i]); }
        g.n = n; g.nA = nA;
        hipError_t e;
        {
            Bracket br(ctx, KID_CONTRACT, bytes);
            e = launch_contract_group(g, cls, ctx->cur);
        }
        if (e == hipSuccess) return AEFFT_OK;
        if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "contract(group)", e);
        (void)hipGetLastError();
    }
    for (int i = 0; i < n; ++i) RET_IF(do_contract(ctx, qs[i]));
    return AEFFT_OK;
}

// pool_fft(conv_k(X)) without the full-resolution conv output (fft_backproplib.cu:1346-1348 when only the pooled
// layer is consumed): Xs[b][r] on the [Nxs][Nys/2+1] grid.  Returns AEFFT_EUNSUPPORTED-like -1 when the kernel declines.
static int do_conv_pooled(aefft_ctx* ctx, const float2* X, const float2* W, const float* bias, float2* Xs, int B, int R, int K,
                          int Nx, int Ny, int Nxs, int Nys, bool* done)
{
    *done = false;
    const long P = bins(Nx, Ny), Ps = bins(Nxs, Nys);
    Contract q{};
    q.A = W; q.a_r = (long)K * P; q.a_k = P;
    q.B = X; q.b_k = P; q.b_c = (long)K * P;
    q.Out = Xs; q.o_r = Ps; q.o_c = (long)R * Ps;
    q.R = R; q.C = B; q.K = K; q.P = Ps;
    q.preDivB = (float)R;
    q.bias = bias; q.biasScale = (float)Nx * (float)Ny; q.biasAfterFirst = true;
    q.gdNx = Nx; q.gdNy = Ny; q.gdNxs = Nxs; q.gdNys = Nys; q.gdMask = 3;
    q = bc(ctx, q);
    hipError_t e;
    {
        Bracket br(ctx, KID_CONTRACT, ((double)R * K + (double)K * B + (double)R * B) * Ps * 8.0);
        e = launch_contract(q, ctx->cur);
    }
    if (e == hipSuccess) { *done = true; return AEFFT_OK; }
    if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "contract(pooled)", e);
    (void)hipGetLastError();
    return AEFFT_OK;
}

// conv_k over a batch: O[b][r] = sum_k (X[b][k]/R) * W[r][k] (+ bias[r]*Nx*Ny at DC)
static int do_conv(aefft_ctx* ctx, const float2* X, const float2* W, const float* bias, float2* O, int B, int R, int K, int Nx, int Ny,
                   float2* Ocrop = nullptr, int Nxs = 0, int Nys = 0)
{
    Contract q = mk_conv(X, W, bias, O, B, R, K, Nx, Ny);
    if (Ocrop) { q.Out2 = Ocrop; q.dnNx = Nx; q.dnNy = Ny; q.dnNxs = Nxs; q.dnNys = Nys; }
    return do_contract(ctx, q);
}

// conv_k whose input is the zero-pad up-sampling (pool_fft with negative scale, fft_backproplib.cu:1360 of the
// previous decoder) of Xs [B][K][sNx][sNy/2+1]: the up-sampled tensor is never materialised.
static int do_conv_up(aefft_ctx* ctx, const float2* Xs, const float2* W, const float* bias, float2* O, int B, int R, int K,
                      int Nx, int Ny, int sNx, int sNy)
{
    if (sNx == Nx && sNy == Ny) return do_conv(ctx, Xs, W, bias, O, B, R, K, Nx, Ny);
    const long P = bins(Nx, Ny), Ps = bins(sNx, sNy);
    Contract q{};
    q.A = W; q.a_r = (long)K * P; q.a_k = P;
    q.B = Xs; q.b_k = Ps; q.b_c = (long)K * Ps;
    q.Out = O; q.o_r = P; q.o_c = (long)R * P;
    q.R = R; q.C = B; q.K = K; q.P = P;
    q.preDivB = (float)R;
    q.bias = bias; q.biasScale = (float)Nx * (float)Ny; q.biasAfterFirst = true;
    q.upNx = Nx; q.upNy = Ny; q.upNxs = sNx; q.upNys = sNy;
    q = bc(ctx, q);
    // algorithmic bytes: the SMALL input, the weights on the support, the full output
    const double bytes = ((double)K * B * Ps + (double)R * K * Ps + (double)R * B * P) * 8.0;
    Bracket br(ctx, KID_CONTRACT, bytes);
    hipError_t e = launch_contract(q, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "contract(up)", e);
    return AEFFT_OK;
}

static int do_resize(aefft_ctx* ctx, const float2* in, float2* out, long planes, int Nx, int Ny, int Nxs, int Nys)
{
    Bracket br(ctx, KID_RESIZE, (double)planes * (std::min(bins(Nx, Ny), bins(Nxs, Nys)) + bins(Nxs, Nys)) * 8.0);
    hipError_t e = launch_resize(in, out, planes, Nx, Ny, Nxs, Nys, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "resize", e);
    return AEFFT_OK;
}

static void pooled(int Nx, int Ny, int scale, int* Nxs, int* Nys)
{
    // fft_backproplib.cu:980-984 with power-of-two scales (exact in float)
    if (scale > 0) { *Nxs = Nx / scale; *Nys = Ny / scale; }
    else { *Nxs = Nx * (-scale); *Nys = Ny * (-scale); }
}

// E = O - T (optional), mse (optional, ACCUMULATED into *mse: caller zeroes), es (optional, accumulated),
// mean over B:  scale = 1/(2*dM*Nx*Ny*B)
static int do_diff_mse(aefft_ctx* ctx, const float2* T, const float2* O, float2* E, float* mse, float* es, int B, int dM, int dD, int Nx, int Ny)
{
    const float scale = 1.0f / ((float)(2 * dM) * (float)Nx * (float)Ny * (float)B);
    Bracket br(ctx, KID_DIFFMSE, (double)B * dD * bins(Nx, Ny) * 8.0 * (E ? 3 : 2));
    hipError_t e = launch_diff_mse(T, O, E, mse, es, B, dD, Nx, Ny, scale, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "diff_mse", e);
    return AEFFT_OK;
}

// gradient_k_io over a batch (fft_backproplib.cu:395-475); T = spectrum of the expected output, E = O - T.
// The reference's four per-bin sums are re-associated so that the batch is contracted FIRST:
//     S[d][d1]  = sum_b  (O_b[d] - T_b[d]) * conj(X_b[d1])            (dD x dD per bin; the subtraction is fused)
//     dc[m][d]  = sum_d1 conj(F[d1][m]) * S[d1][d]      / (Norm*B)   (== conj(X) * sum_d1 E conj(F), :421-439)
//     df[d][m]  = sum_d1 S[d][d1] * conj(C[m][d1])      / (Norm*B)   (== E * conj(sum_d1 C X),      :426-455)
//     df[d][m](0,0) += es[d] * b[m]*Nx*Ny / (Norm*B),  es[d] = sum_b E_b[d](0,0)   (the b0 term, :448-455)
// Same sums, different order (float32 rounding only); neither E nor the B*dM-plane intermediates of the
// literal form are materialised.  S: workspace [dD][dD][P].  dc and df are produced by ONE launch.
static int do_gradient(aefft_ctx* ctx, const float2* Xin, const float2* T, const float2* O, const float2* C, const float2* F,
                       const float* b, float2* S, float2* dc, float2* df, float* db, float* dp, int B, int dM, int dD, int Nx, int Ny)
{
    const long P = bins(Nx, Ny);
    const float norm = (float)Nx * (float)Ny;                 // fft_backproplib.cu:398
    const float Norm = grad_norm(dM, dD, Nx, Ny);
    RET_IF(do_contract(ctx, mk_S(Xin, T, O, S, B, dD, P)));
    RET_IF(do_contract2(ctx, mk_dc(F, S, dc, B, dM, dD, P, Norm), mk_df(C, S, df, B, dM, dD, P, Norm)));
    {
        Bracket br(ctx, KID_BIASGRAD, ((double)(dM * dD + dM + dD) + 2.0 * B * dD) * 8.0);
        hipError_t e = launch_bias_grad(O, T, F, b, df, db, dp, B, dM, dD, P, norm, Norm, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "bias_grad", e);
    }
    return AEFFT_OK;
}

// unnormalised C2R of a gradient spectrum sampled on the kernel support: g[planes][Nk][Nl]
// (== shrink_k(cufftExecC2R(d)), fft_backproplib.cu:1219-1226).  Direct pruned evaluation when the
// support is 3x3/5x5/7x7, generic C2R + shrink otherwise.
static int do_c2r_shrink(aefft_ctx* ctx, const float2* dspec, float* gk, float* realws, float* part, long planes, int Nx, int Ny, int Nk, int Nl, float scale = 1.0f)
{
    if (pruned_supported(Nk, Nl, Nx, Ny)) {
        Bracket br(ctx, KID_KGRAD, (double)planes * (bins(Nx, Ny) * 8.0 + Nk * Nl * 4.0));
        hipError_t e = launch_kgrad(dspec, gk, part, ctx->tw, planes, Nx, Ny, Nk, Nl, scale, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "kgrad", e);
        return AEFFT_OK;
    }
    RET_IF(do_c2r(ctx, dspec, realws, planes, Nx, Ny, Nx, Ny, scale));
    Bracket br(ctx, KID_SHRINK, (double)planes * Nk * Nl * 8.0);
    hipError_t e = launch_shrink(realws, gk, planes, Nx, Ny, Nk, Nl, 1.0f, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "shrink", e);
    return AEFFT_OK;
}

// pad + R2C: kernel [planes][Nk][Nl] -> spectrum [planes][Nx][Nyr]  (fft_backproplib.cu:1274-1282 / 1150-1152)
static int do_pad_r2c(aefft_ctx* ctx, const float* k, float2* K, float* realws, long planes, int Nx, int Ny, int Nk, int Nl)
{
    if (pruned_supported(Nk, Nl, Nx, Ny)) {
        Bracket br(ctx, KID_KSPEC, (double)planes * (bins(Nx, Ny) * 8.0 + Nk * Nl * 4.0));
        hipError_t e = launch_kspec(k, K, ctx->tw, planes, Nx, Ny, Nk, Nl, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "kspec", e);
        return AEFFT_OK;
    }
    {
        Bracket br(ctx, KID_PAD, (double)planes * ((double)Nx * Ny + Nk * Nl) * 4.0);
        hipError_t e = launch_pad(k, realws, planes, Nx, Ny, Nk, Nl, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "pad", e);
    }
    return do_r2c(ctx, realws, K, planes, Nx, Ny, Nx, Ny);
}

struct Momentum { float *Dc, *Df, *Db, *Dp; };

// the coordinate-space part of `backprop` (fft_backproplib.cu:1229-1272) on already shrunk gradients
static int do_update(aefft_ctx* ctx, float* c, float* f, float* b, float* p, const float* dck, const float* dfk, const float* db,
                     const float* dp, Momentum mo, int dM, int dD, int Nk, int Nl, float del, int maxdiff, int sym, float gscale,
                     float* zero = nullptr)
{
    UpdateArgs a{};
    a.zero = zero;
    a.c = c; a.f = f; a.b = b; a.p = p;
    a.dck = dck; a.dfk = dfk; a.db = db; a.dp = dp;
    a.Dc = mo.Dc; a.Df = mo.Df; a.Db = mo.Db; a.Dp = mo.Dp;
    a.dM = dM; a.dD = dD; a.Nk = Nk; a.Nl = Nl;
    a.del = del; a.alpha = 0.9f; a.w0 = 1.f; a.w1 = 10.f;      // fft_backproplib.cu:608,1252
    a.gscale = sym ? 0.5f * gscale : gscale; a.sym = sym;
    if (maxdiff) {
        const size_t nk = (size_t)dM * dD * Nk * Nl;
        void *small, *den;
        RET_IF(ws_get(ctx, WS_SMALL, sizeof(float) * (2 * nk + dM + dD + 64), &small));
        RET_IF(ws_get(ctx, WS_DEN, sizeof(float) * gradient_diff_ws_floats(dM, dD, Nk, Nl), &den));
        float* cd = (float*)small; float* fd = cd + nk; float* bd = fd + nk; float* pd = bd + dM;
        {
            Bracket br(ctx, KID_GDIFF, (double)nk * 16.0);
            hipError_t e = launch_gradient_diff(c, f, b, p, cd, fd, bd, pd, (float*)den, dM, dD, Nk, Nl, ctx->cur);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "gradient_diff", e);
        }
        a.cd = cd; a.fd = fd; a.bd = bd; a.pd = pd;
    }
    Bracket br(ctx, KID_UPDATE, (double)dM * dD * Nk * Nl * 4.0 * 8);
    hipError_t e = launch_update(a, ctx->cur);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "update", e);
    return AEFFT_OK;
}

static UpdateArgs mk_update(float* c, float* f, float* b, float* p, const float* dck, const float* dfk, const float* db, const float* dp,
                            Momentum mo, int dM, int dD, int Nk, int Nl, float del, int sym, float gscale, float* zero)
{
    UpdateArgs a{};
    a.zero = zero;
    a.c = c; a.f = f; a.b = b; a.p = p;
    a.dck = dck; a.dfk = dfk; a.db = db; a.dp = dp;
    a.Dc = mo.Dc; a.Df = mo.Df; a.Db = mo.Db; a.Dp = mo.Dp;
    a.dM = dM; a.dD = dD; a.Nk = Nk; a.Nl = Nl;
    a.del = del; a.alpha = 0.9f; a.w0 = 1.f; a.w1 = 10.f;      // fft_backproplib.cu:608,1252
    a.gscale = sym ? 0.5f * gscale : gscale; a.sym = sym;
    return a;
}

// ------------------------------------------------------------------------------------------
// op-level C entry points
// ------------------------------------------------------------------------------------------
#define CF2(p) reinterpret_cast<const float2*>(p)
#define F2(p) reinterpret_cast<float2*>(p)

extern "C" int aefft_r2c(aefft_ctx* ctx, const float* x_d, float* X_d, long planes, int Nx, int Ny)
{
    if (!ctx || !x_d || !X_d || planes < 0) return fail(ctx, AEFFT_EINVAL, "aefft_r2c: bad argument");
    return do_r2c(ctx, x_d, F2(X_d), planes, Nx, Ny, Nx, Ny);
}

extern "C" int aefft_c2r(aefft_ctx* ctx, const float* X_d, float* x_d, long planes, int Nx, int Ny, float scale)
{
    if (!ctx || !x_d || !X_d || planes < 0) return fail(ctx, AEFFT_EINVAL, "aefft_c2r: bad argument");
    return do_c2r(ctx, CF2(X_d), x_d, planes, Nx, Ny, Nx, Ny, scale);
}

// op level: any integer scale, sized as the reference sizes it (fft_backproplib.cu:980-984: l = scale or 1/|scale| as FLOAT, Nxs = int(Nx / l) --
// exact for powers of two, SURVEY B-4, and for the other scales whatever that float arithmetic gives); the resized grid must be even (the
// index rules of `resize`, :98-153, are written for even sizes) and a size the transforms serve
static int chk_scale(aefft_ctx* ctx, int Nx, int Ny, int scale, int* Nxs, int* Nys)
{
    if (scale == 0) return fail(ctx, AEFFT_EINVAL, "pooling scale must be non-zero");
    const float l = scale > 0 ? (float)scale : -1.0f / (float)scale;
    *Nxs = (int)((float)Nx / l); *Nys = (int)((float)Ny / l);
    if ((*Nxs & 1) || (*Nys & 1)) return fail(ctx, AEFFT_EINVAL, "pooled size must be even");
    if (*Nxs < 8 || *Nys < 8 || *Nxs > 2048 || *Nys > 2048) return fail(ctx, AEFFT_EINVAL, "pooled size out of range 8..2048");
    return AEFFT_OK;
}

extern "C" int aefft_pool(aefft_ctx* ctx, const float* X_d, float* Xs_d, long planes, int Nx, int Ny, int scale, int* Nxs, int* Nys)
{
    if (!ctx || !X_d || !Xs_d || planes < 0) return fail(ctx, AEFFT_EINVAL, "aefft_pool: bad argument");
    RET_IF(chk_size_any(ctx, Nx, Ny));
    int nx, ny;
    RET_IF(chk_scale(ctx, Nx, Ny, scale, &nx, &ny));
    if (Nxs) *Nxs = nx;
    if (Nys) *Nys = ny;
    if (scale == 1 || scale == -1) {   // fft_backproplib.cu:977: nothing happens
        HIPCHK(ctx, hipMemcpyAsync(Xs_d, X_d, sizeof(float2) * planes * bins(Nx, Ny), hipMemcpyDeviceToDevice, ctx->stream));
        return AEFFT_OK;
    }
    return do_resize(ctx, CF2(X_d), F2(Xs_d), planes, Nx, Ny, nx, ny);
}

extern "C" int aefft_r2c_pool(aefft_ctx* ctx, const float* x_d, float* Xs_d, long planes, int Nx, int Ny, int scale)
{
    if (!ctx || !x_d || !Xs_d || planes < 0 || scale < 1) return fail(ctx, AEFFT_EINVAL, "aefft_r2c_pool: bad argument");
    int nx, ny;
    RET_IF(chk_scale(ctx, Nx, Ny, scale, &nx, &ny));
    return do_r2c(ctx, x_d, F2(Xs_d), planes, Nx, Ny, nx, ny);
}

extern "C" int aefft_unpool_c2r(aefft_ctx* ctx, const float* Xs_d, float* x_d, long planes, int Nxs, int Nys, int scale, float out_scale)
{
    if (!ctx || !x_d || !Xs_d || planes < 0 || scale > -1) return fail(ctx, AEFFT_EINVAL, "aefft_unpool_c2r: scale must be <= -1");
    int nx, ny;
    RET_IF(chk_scale(ctx, Nxs, Nys, scale, &nx, &ny));
    return do_c2r(ctx, CF2(Xs_d), x_d, planes, Nxs, Nys, nx, ny, out_scale);
}

extern "C" int aefft_kernel_spectrum(aefft_ctx* ctx, const float* k_d, float* K_d, int nA, int nB, int Nk, int Nl, int Nx, int Ny)
{
    if (!ctx || !k_d || !K_d || nA <= 0 || nB <= 0 || Nk <= 0 || Nl <= 0 || Nk > Nx || Nl > Ny) return fail(ctx, AEFFT_EINVAL, "aefft_kernel_spectrum: bad argument");
    RET_IF(chk_size(ctx, Nx, Ny));
    const long planes = (long)nA * nB;
    void* real = nullptr;
    if (!pruned_supported(Nk, Nl, Nx, Ny)) RET_IF(ws_get(ctx, WS_REAL, sizeof(float) * planes * Nx * Ny, &real));
    return do_pad_r2c(ctx, k_d, F2(K_d), (float*)real, planes, Nx, Ny, Nk, Nl);
}

extern "C" int aefft_kernel_export(aefft_ctx* ctx, const float* K_d, float* k_d, int nA, int nB, int Nk, int Nl, int Nx, int Ny)
{
    if (!ctx || !k_d || !K_d || nA <= 0 || nB <= 0 || Nk <= 0 || Nl <= 0 || Nk > Nx || Nl > Ny) return fail(ctx, AEFFT_EINVAL, "aefft_kernel_export: bad argument");
    RET_IF(chk_size(ctx, Nx, Ny));
    const long planes = (long)nA * nB;
    void *real = nullptr, *part = nullptr;
    if (pruned_supported(Nk, Nl, Nx, Ny)) RET_IF(ws_get(ctx, WS_PART, sizeof(float) * kgrad_partial_floats(planes, Nx, Ny, Nk, Nl), &part));
    else RET_IF(ws_get(ctx, WS_REAL, sizeof(float) * planes * Nx * Ny, &real));
    // kfft_inv: C2R then * 1/(Nx*Ny) (fft_backproplib.cu:948), then kernel_invpad
    return do_c2r_shrink(ctx, CF2(K_d), k_d, (float*)real, (float*)part, planes, Nx, Ny, Nk, Nl, 1.0f / ((float)Nx * (float)Ny));
}

extern "C" int aefft_conv(aefft_ctx* ctx, const float* X_d, const float* C_d, const float* bias_d, float* O_d, int B, int dM, int dD, int Nx, int Ny)
{
    if (!ctx || !X_d || !C_d || !O_d || B <= 0 || dM <= 0 || dD <= 0) return fail(ctx, AEFFT_EINVAL, "aefft_conv: bad argument");
    RET_IF(chk_size(ctx, Nx, Ny));
    if (!aligned16(X_d) || !aligned16(C_d) || !aligned16(O_d)) return fail(ctx, AEFFT_EINVAL, "aefft_conv: pointers must be 16-byte aligned");
    return do_conv(ctx, CF2(X_d), CF2(C_d), bias_d, F2(O_d), B, dM, dD, Nx, Ny);
}

extern "C" int aefft_gradient(aefft_ctx* ctx, const float* Xin_d, const float* Xout_d, const float* O_d, const float* C_d,
                              const float* F_d, const float* b_d, float* dc_d, float* df_d, float* db_d, float* dp_d,
                              int B, int dM, int dD, int Nx, int Ny)
{
    if (!ctx || !Xin_d || !Xout_d || !O_d || !C_d || !F_d || !b_d || !dc_d || !df_d || !db_d || !dp_d || B <= 0 || dM <= 0 || dD <= 0)
        return fail(ctx, AEFFT_EINVAL, "aefft_gradient: bad argument");
    RET_IF(chk_size(ctx, Nx, Ny));
    const long P = bins(Nx, Ny);
    void* S;
    RET_IF(ws_get(ctx, WS_S, sizeof(float2) * dD * dD * P, &S));
    return do_gradient(ctx, CF2(Xin_d), CF2(Xout_d), CF2(O_d), CF2(C_d), CF2(F_d), b_d, (float2*)S, F2(dc_d), F2(df_d), db_d, dp_d, B, dM, dD, Nx, Ny);
}

extern "C" int aefft_mse(aefft_ctx* ctx, const float* T_d, const float* O_d, float* mse_d, int B, int dM, int dD, int Nx, int Ny)
{
    if (!ctx || !T_d || !O_d || !mse_d || B <= 0) return fail(ctx, AEFFT_EINVAL, "aefft_mse: bad argument");
    HIPCHK(ctx, hipMemsetAsync(mse_d, 0, sizeof(float), ctx->stream));
    return do_diff_mse(ctx, CF2(T_d), CF2(O_d), nullptr, mse_d, nullptr, B, dM, dD, Nx, Ny);
}

extern "C" int aefft_update(aefft_ctx* ctx, float* c_d, float* f_d, float* b_d, float* p_d, float* C_d, float* F_d,
                            const float* dc_d, const float* df_d, const float* db_d, const float* dp_d,
                            float* Dc_d, float* Df_d, float* Db_d, float* Dp_d,
                            int dM, int dD, int Nx, int Ny, int Nk, int Nl, float del, int maxdiff)
{
    if (!ctx || !c_d || !f_d || !b_d || !p_d || !C_d || !F_d || !dc_d || !df_d || !db_d || !dp_d || !Dc_d || !Df_d || !Db_d || !Dp_d)
        return fail(ctx, AEFFT_EINVAL, "aefft_update: null pointer");
    RET_IF(chk_size(ctx, Nx, Ny));
    const long planes = (long)dM * dD;
    const size_t nk = (size_t)planes * Nk * Nl;
    void *real = nullptr, *tmp, *part = nullptr;
    if (pruned_supported(Nk, Nl, Nx, Ny)) RET_IF(ws_get(ctx, WS_PART, sizeof(float) * kgrad_partial_floats(planes, Nx, Ny, Nk, Nl), &part));
    else RET_IF(ws_get(ctx, WS_REAL, sizeof(float) * planes * Nx * Ny, &real));
    RET_IF(ws_get(ctx, WS_TMP, sizeof(float) * 2 * nk, &tmp));
    float* dck = (float*)tmp; float* dfk = dck + nk;
    RET_IF(do_c2r_shrink(ctx, CF2(dc_d), dck, (float*)real, (float*)part, planes, Nx, Ny, Nk, Nl));
    RET_IF(do_c2r_shrink(ctx, CF2(df_d), dfk, (float*)real, (float*)part, planes, Nx, Ny, Nk, Nl));
    RET_IF(do_update(ctx, c_d, f_d, b_d, p_d, dck, dfk, db_d, dp_d, Momentum{Dc_d, Df_d, Db_d, Dp_d}, dM, dD, Nk, Nl, del, maxdiff, 0, 1.0f));
    RET_IF(do_pad_r2c(ctx, c_d, F2(C_d), (float*)real, planes, Nx, Ny, Nk, Nl));
    RET_IF(do_pad_r2c(ctx, f_d, F2(F_d), (float*)real, planes, Nx, Ny, Nk, Nl));
    return AEFFT_OK;
}

// spatial mode ------------------------------------------------------------------------------
static void spatial_geom(int Nk, int Nl, int cpu_semantics, int* ak, int* al, int* lo)
{
    if (cpu_semantics == 1) { *ak = (Nk - 1) / 2 - 1; *al = (Nl - 1) / 2 - 1; *lo = 1; }     // netlib.cpp:325-326,344
    else { *ak = ((Nk - 1) / 2 - 1) / 2; *al = ((Nl - 1) / 2 - 1) / 2; *lo = 0; }             // backproplib.cu:123-124,95
}

extern "C" int aefft_conv_spatial(aefft_ctx* ctx, const float* in_d, float* out_d, const float* c_d, const float* b_d,
                                  int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl, int cpu_semantics)
{
    if (!ctx || !in_d || !out_d || !c_d || !b_d || B <= 0 || dD <= 0 || dM <= 0 || Nx <= 0 || Ny <= 0 || Nk <= 0 || Nl <= 0)
        return fail(ctx, AEFFT_EINVAL, "aefft_conv_spatial: bad argument");
    int ak, al, lo;
    spatial_geom(Nk, Nl, cpu_semantics, &ak, &al, &lo);
    Bracket br(ctx, KID_SPATIAL, ((double)B * (dD + dM) * Nx * Ny + (double)dM * dD * Nk * Nl) * 4.0);
    hipError_t e = launch_conv_spatial(in_d, out_d, c_d, b_d, B, dD, dM, Nx, Ny, Nk, Nl, ak, al, cpu_semantics == 1 ? 1.f : (float)dM, lo, ctx->stream);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "conv_spatial", e);
    return AEFFT_OK;
}

extern "C" int aefft_pool_conv_spatial(aefft_ctx* ctx, const float* in_d, float* pooled_d, float* out_d, const float* c_d, const float* b_d,
                                       int B, int dD, int dM, int Nx, int Ny, int scale, int Nk, int Nl, int cpu_semantics)
{
    if (!ctx || !in_d || !out_d || !c_d || !b_d || B <= 0 || dD <= 0 || dM <= 0 || Nx <= 0 || Ny <= 0 || Nk <= 0 || Nl <= 0 || scale < 1)
        return fail(ctx, AEFFT_EINVAL, "aefft_pool_conv_spatial: bad argument");
    int ak, al, lo;
    spatial_geom(Nk, Nl, cpu_semantics, &ak, &al, &lo);
    Bracket br(ctx, KID_SPATIAL, ((double)B * dD * Nx * Ny * scale * scale + (double)B * (dM + (pooled_d ? dD : 0)) * Nx * Ny + (double)dM * dD * Nk * Nl) * 4.0);
    hipError_t e = launch_conv_spatial(in_d, out_d, c_d, b_d, B, dD, dM, Nx, Ny, Nk, Nl, ak, al, cpu_semantics == 1 ? 1.f : (float)dM, lo, ctx->stream, scale, pooled_d);
    if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return fail(ctx, AEFFT_EINVAL, "aefft_pool_conv_spatial: kernel shape not served by the fused kernel"); }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "pool_conv_spatial", e);
    return AEFFT_OK;
}

extern "C" int aefft_pool_spatial(aefft_ctx* ctx, const float* in_d, float* out_d, long planes, int Nxi, int Nyi, int Nxo, int Nyo, int scale)
{
    if (!ctx || !in_d || !out_d || planes <= 0 || Nxi <= 0 || Nyi <= 0 || Nxo <= 0 || Nyo <= 0 || scale == 0)
        return fail(ctx, AEFFT_EINVAL, "aefft_pool_spatial: bad argument");
    Bracket br(ctx, KID_SPATIAL, (double)planes * ((double)Nxi * Nyi + (double)Nxo * Nyo) * 4.0);
    hipError_t e = launch_pool_spatial(in_d, out_d, planes, Nxi, Nyi, Nxo, Nyo, scale, ctx->stream);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "pool_spatial", e);
    return AEFFT_OK;
}

static int backprop_spatial_impl(aefft_ctx* ctx, const float* in_d, const float* out_d, const float* hin_d,
                                 float* c_d, float* b_d, float* f_d, float* p_d,
                                 float* dc_d, float* db_d, float* df_d, float* dp_d,
                                 float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                                 int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                                 float delmax, float alpha, int tied, int cpu_semantics, bool hin_is_conv);

extern "C" int aefft_backprop_spatial(aefft_ctx* ctx, const float* in_d, const float* out_d, const float* hin_d,
                                      float* c_d, float* b_d, float* f_d, float* p_d,
                                      float* dc_d, float* db_d, float* df_d, float* dp_d,
                                      float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                                      int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                                      float delmax, float alpha, int tied, int cpu_semantics)
{
    return backprop_spatial_impl(ctx, in_d, out_d, hin_d, c_d, b_d, f_d, p_d, dc_d, db_d, df_d, dp_d, ddc_d, ddb_d, ddf_d, ddp_d,
                                 B, dD, dM, Nx, Ny, Nk, Nl, delmax, alpha, tied, cpu_semantics, false);
}

extern "C" int aefft_step_spatial(aefft_ctx* ctx, const float* in_d, float* hin_d, float* out_d,
                                  float* c_d, float* b_d, float* f_d, float* p_d,
                                  float* dc_d, float* db_d, float* df_d, float* dp_d,
                                  float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                                  int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                                  float delmax, float alpha, int tied, int cpu_semantics)
{
    if (!ctx || !in_d || !out_d || !hin_d || cpu_semantics < 0 || cpu_semantics > 1) return fail(ctx, AEFFT_EINVAL, "aefft_step_spatial: bad argument");
    RET_IF(aefft_conv_spatial(ctx, in_d, hin_d, c_d, b_d, B, dD, dM, Nx, Ny, Nk, Nl, cpu_semantics));
    RET_IF(aefft_conv_spatial(ctx, hin_d, out_d, f_d, p_d, B, dM, dD, Nx, Ny, Nk, Nl, cpu_semantics));
    return backprop_spatial_impl(ctx, in_d, out_d, hin_d, c_d, b_d, f_d, p_d, dc_d, db_d, df_d, dp_d, ddc_d, ddb_d, ddf_d, ddp_d,
                                 B, dD, dM, Nx, Ny, Nk, Nl, delmax, alpha, tied, cpu_semantics, true);
}

static int backprop_spatial_impl(aefft_ctx* ctx, const float* in_d, const float* out_d, const float* hin_d,
                                 float* c_d, float* b_d, float* f_d, float* p_d,
                                 float* dc_d, float* db_d, float* df_d, float* dp_d,
                                 float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                                 int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                                 float delmax, float alpha, int tied, int cpu_semantics, bool hin_is_conv)
{
    if (!ctx || !in_d || !out_d || !hin_d || !c_d || !b_d || !f_d || !p_d || !dc_d || !db_d || !df_d || !dp_d || B <= 0)
        return fail(ctx, AEFFT_EINVAL, "aefft_backprop_spatial: bad argument");
    const size_t nk = (size_t)dM * dD * Nk * Nl;
    void *ws, *small;
    RET_IF(ws_get(ctx, WS_REAL, sizeof(float) * (size_t)B * dM * Nx * Ny, &ws));
    const size_t nrq = spatial_rq_floats(dD, Nk, Nl);
    RET_IF(ws_get(ctx, WS_TMP, sizeof(float) * (2 * nk + dM + dD + nrq), &small));
    SpatialGradArgs a{};
    a.in = in_d; a.out = out_d; a.hin = hin_d; a.f = f_d;
    a.gc = (float*)small; a.gf = a.gc + nk; a.gb = a.gf + nk; a.gp = a.gb + dM;
    a.ws = (float*)ws;
    a.rq = a.gp + dD;
    {
        const size_t pf = spatial_partial_floats(B, dD, dM, Nx, Ny, Nk, Nl);
        void* part = nullptr;
        if (pf) RET_IF(ws_get(ctx, WS_PART, sizeof(float) * pf, &part));
        a.part = (float*)part;
    }
    a.B = B; a.dD = dD; a.dM = dM; a.Nx = Nx; a.Ny = Ny; a.Nk = Nk; a.Nl = Nl;
    spatial_geom(Nk, Nl, cpu_semantics, &a.ak, &a.al, &a.lo);
    a.Norm = (float)(dD * dM * Nk * Nl * Nx * Ny);            // backproplib.cu:303
    if (tied) a.Norm = (float)(2 * dD * dM * Nk * Nl * Nx * Ny);   // :533
    a.tied = tied;
    if (hin_is_conv && spatial_regions_ok(a)) {
        // the hidden layer is this call's own Conv_gpu(in; c, b): dF and dP come out of the error-input region sums as dC and dB do, the
        // hidden layer is not read again (the weights are read before the update below changes them: stream order)
        a.c1 = c_d; a.b1 = b_d; a.div1 = cpu_semantics == 1 ? 1.f : (float)dM;
    }
    {
        Bracket br(ctx, KID_SPATIAL, (double)B * ((a.c1 ? 2.0 : 3.0) * dD + (a.c1 ? 0.0 : 1.0) * dM) * Nx * Ny * 4.0);
        hipError_t e = launch_spatial_grad(a, ctx->stream);
        if (e == hipSuccess && cpu_semantics == 2) e = launch_spatial_compat(a, ctx->stream);      // Appendix B-11: bug-compatible gf, gb
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "spatial_grad", e);
    }
    UpdateArgs u{};
    u.c = c_d; u.f = f_d; u.b = b_d; u.p = p_d;
    u.dck = a.gc; u.dfk = a.gf; u.db = a.gb; u.dp = a.gp;
    u.Dc = dc_d; u.Df = df_d; u.Db = db_d; u.Dp = dp_d;
    u.dM = dM; u.dD = dD; u.Nk = Nk; u.Nl = Nl;
    u.del = delmax; u.alpha = alpha; u.w0 = 1.f; u.w1 = 0.f; u.gscale = 1.f; u.sym = tied;
    u.ddc = ddc_d; u.ddf = tied ? nullptr : ddf_d; u.ddb = ddb_d; u.ddp = ddp_d;   // adapt_rate records the gradient (backproplib.cu:33)
    {
        Bracket br(ctx, KID_UPDATE, (double)nk * 4.0 * 8);
        hipError_t e = launch_update(u, ctx->stream);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "update", e);
    }
    return AEFFT_OK;
}

// ------------------------------------------------------------------------------------------
// resident network
// ------------------------------------------------------------------------------------------
struct Pair {
    int dD, dM, Nk, Nl, s;
    int Nxin, Nyin;          // resolution before this pair's pooling
    int Nx, Ny;              // working resolution (after pooling)
    long P;
    float *c, *f, *b, *p;
    float *Dc, *Df, *Db, *Dp;
    float2 *C, *F;
    bool spectra_valid;
    bool H_stale = false;    // the last (lazy) forward produced only the pooled part of H: recompute before reading H
    float2* G = nullptr;     // [dD][dD][P] collapsed pair operator F.C/(dM dD) (post-update MSE; innermost pair's forward)
    bool G_valid = false;    // G and beta (its DC bias) belong to the CURRENT weights (left by aefft_net_step_apply)
    float* beta = nullptr;   // [dD]
    float* Q = nullptr;      // [dD][dD][Qn][T*T], T = 2Nk-1: pruned inverse transform of S (weight_kernels.hip), Qn row-chunk partial sums
    int Qn = 1;
    float2* Oc = nullptr;    // [B][dD][Pc] decoder output on the support of the up-sampled spectra (the coarsest pair's grid); last pair: == O
    float2* opA[2] = {nullptr, nullptr};   // [OPC][dD][P]  operator chain: the pair's input on the basis frames, two sets (the step in progress / the next step's)
    float2* opO[2] = {nullptr, nullptr};   // [OPC][dD][Pc] ... its decoder output on the coarsest grid's support
    float2* Cc = nullptr;    // [dM][dD][P of the next pair] C sampled where the next pair's grid lands (operator chain: its planar tiles read nothing else of C)
    bool O_stale = false;    // the last (lazy) forward produced Oc only: expand before reading O
    float2 *X, *H, *O;       // [B][dD][P], [B][dM][P], [B][dD][P] (X aliases the previous pair's H when s == 1)
    size_t goff;             // offset (floats) of this pair's segment in the packed gradient buffer
    float* es;               // [2*dD] DC bins of the error summed over the batch (inside the net scratch)
    float2 *S, *dc, *df;     // per-pair gradient workspaces (pairs run concurrently on side streams); df == dc + W
    float* part;             // kgrad partial sums
};

struct aefft_net {
    aefft_ctx* ctx;
    int D, Nx, Ny, L, B;
    std::vector<Pair> pr;
    std::vector<void*> allocs;
    // operator form of the training step (opform_kernels.hip)
    int Bc = 0;                // columns every activation buffer is allocated for: max(B, OPC)
    float2* Xf = nullptr;      // [B][D][P0] input spectra of the frames (pair 0's X in the per-frame form)
    float2* A0hat = nullptr;   // [OPC][D][P0] basis frames (pair 0's X in the operator form); null: D > OPC-1
    float2* Mhat = nullptr;    // [OPC][OPC][P0] second moments of the batch
    bool op_state = false;     // the activation buffers hold OPERATORS (basis-frame responses) of the last step_grad, not frames
    // chain mode (the step's forward is chain_kernel): the operators live in their OWN buffers (Pair::opA / opO), two sets, because the
    // tail launch of a step already runs the NEXT step's chain on the updated weights (it depends on the weights only) while the
    // post-update MSE still reads this step's operators
    bool op_chain = false;     // the last step_grad ran in chain mode: operators in set op_fwd, activation buffers NOT refreshed (act_stale)
    bool act_stale = false;    // the activation buffers do not hold the last forward's frames (ensure_frames expands them from the operators)
    bool upd_after_fwd = false; // aefft_net_step_apply has changed the weights since the step's forward: a layer export forms a skipped hidden layer with
                               // the encoder of THAT forward, recovered as w + D (the momentum buffer holds the step that was applied)
    bool chain_valid = false;  // set op_set holds the operators of the CURRENT weights
    int op_set = 0, op_fwd = 0;
    float2* Wp = nullptr;      // [Pc][packE] bin-major copy of the kernel spectra the coarsest-grid chain items read (kspec_packed_kernel)
    PackArgs pack{};           // its description (static per net)
    bool packed_valid = false; // Wp belongs to the current weights
    float* grad = nullptr; size_t grad_n = 0;
    float* scratch = nullptr;  // [mse_pre[L] | mse_post[L] | es of pair 0 (2*dD) | es of pair 1 | ...], zeroed once per step
    size_t scratch_n = 0;
    float* mse_pre = nullptr;  // = scratch
    float* mse_post = nullptr; // = scratch + L
    float* gtaps = nullptr;      // G' from the stored taps on HBM-sized grids: the (2Nk-1)^2 taps of every plane of every pair (gprime_from_taps)
    float *gd_out = nullptr, *gd_part = nullptr;   // multiobjective mode: [cd | fd | bd | pd] per pair, and the chunk partial sums (gradient_diff_ws_floats)
    bool mse_pending = false;   // the slots hold the unsummed post-update MSE of the last aefft_net_step_apply (mse_d == NULL): summed by the next step's wgrad launch or mse_flush
    float mse_pending_scale = 1.f;
    float* mse_slots = nullptr; // [L][MSE_SLOTS*MSE_SLOT_STRIDE] accumulators of the fused re-forward MSE (zero between uses)
    float* mse_dev = nullptr;  // scratch for bursts
    size_t mse_cap = 0;
    const float* last_frames = nullptr;
    bool last_frames_u8 = false;     // ... and they were 8-bit pixels
    bool have_forward = false, have_grad = false;
    int NxC = 0, NyC = 0; long Pc = 0;   // grid of the coarsest pair = support of every decoder output
    bool compact = true;                 // the training step may keep decoder outputs on that support only
    // input prefetch (aefft_net_set_input_ready): second buffer for pair 0's input spectra, end-of-step events, step counter
    bool input_ready = false;
    float2* X0alt = nullptr;
    hipEvent_t ev_end[2] = {nullptr, nullptr}, ev_r2c = nullptr, ev_mid = nullptr;
    bool ev_mid_valid = false;
    bool ev_end_valid[2] = {false, false};
    unsigned long step_no = 0;
    float2* recon_exp = nullptr;  // [B][D][PO] per-frame output spectra of the reconstruction when they are written out (large supports, launch_recon)
    unsigned ox_done = 0;         // bit l: the forward already launched pair l's support term S += sum_b Oc X^H
    bool xx_done = false;         // the forward already launched S = -sum_b X X^H (grouped with the innermost decoder conv)
    bool recon_pending = false;   // the reconstruction's inverse FFT is still running on aux[0]
    float* recon_deferred = nullptr;   // pipelined mode: the reconstruction is launched at the end of the gradient half
    bool burst = false;        // inside aefft_net_train_pair (its MSE slots are zeroed up front, not by the update kernel)
    // shared scratch sized for the largest pair
    bool fuse_crop = true;     // encoder convs also write the next pair's cropped input (no resize launches)
    bool pruned = true;        // every pair's kernel support has a pruned transform -> no shared FFT workspace in the backward
    float* real;
};

static int net_alloc(aefft_net* n, void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) return fail(n->ctx, AEFFT_ENOMEM, "hipMalloc(net)", e);
    if (flag(AEFFT_F_POISON)) { (void)hipMemset(*p, 0xFF, std::max<size_t>(bytes, 256)); (void)hipDeviceSynchronize(); }
    n->allocs.push_back(*p);
    return AEFFT_OK;
}
template <typename T> static int net_alloc_t(aefft_net* n, T** p, size_t count) { return net_alloc(n, reinterpret_cast<void**>(p), count * sizeof(T)); }

static int build_chain_items(aefft_net* n);

extern "C" void aefft_net_destroy(aefft_net* net)
{
    if (!net) return;
    (void)hipStreamSynchronize(net->ctx->stream);
    for (int i = 0; i < aefft_ctx::NAUX; ++i) if (net->ctx->aux[i]) (void)hipStreamSynchronize(net->ctx->aux[i]);
    for (int i = 0; i < 2; ++i) if (net->ev_end[i]) (void)hipEventDestroy(net->ev_end[i]);
    if (net->ev_r2c) (void)hipEventDestroy(net->ev_r2c);
    if (net->ev_mid) (void)hipEventDestroy(net->ev_mid);
    for (void* p : net->allocs) (void)hipFree(p);
    delete net;
}

extern "C" int aefft_net_create(aefft_ctx* ctx, const aefft_net_desc* d, aefft_net** out)
{
    if (!ctx || !d || !out || d->npairs <= 0 || d->batch <= 0 || d->D <= 0 || !d->maps || !d->Nk || !d->Nl || !d->scale)
        return fail(ctx, AEFFT_EINVAL, "aefft_net_create: bad descriptor");
    *out = nullptr;
    RET_IF(chk_size(ctx, d->Nx, d->Ny));
    aefft_net* n = new aefft_net();
    n->ctx = ctx; n->D = d->D; n->Nx = d->Nx; n->Ny = d->Ny; n->L = d->npairs; n->B = d->batch;
    n->Bc = std::max(n->B, (int)OPC);
    n->pr.resize(n->L);
    int dD = d->D, nx = d->Nx, ny = d->Ny;
    size_t maxS = 0, maxBDP = 0, maxW = 0, maxReal = 0, goff = 0, maxMid = 0, maxDen = 0, maxSmall = 0, soff = 2 * (size_t)d->npairs;
    std::vector<size_t> esoff(d->npairs);
    int rc = AEFFT_OK;
    for (int l = 0; l < n->L && rc == AEFFT_OK; ++l) {
        Pair& q = n->pr[l];
        q.dD = dD; q.dM = d->maps[l]; q.Nk = d->Nk[l]; q.Nl = d->Nl[l]; q.s = d->scale[l];
        q.Nxin = nx; q.Nyin = ny;
        if (q.dM <= 0 || q.Nk <= 0 || q.Nl <= 0 || q.s < 1 || !pow2(q.s)) { rc = fail(ctx, AEFFT_EINVAL, "aefft_net_create: bad pair parameters"); break; }
        q.Nx = nx / q.s; q.Ny = ny / q.s;
        if (q.Nx < 8 || q.Ny < 8 || q.Nk > q.Nx || q.Nl > q.Ny) { rc = fail(ctx, AEFFT_EINVAL, "aefft_net_create: pooled size < 8 or kernel larger than plane"); break; }
        q.P = bins(q.Nx, q.Ny);
        const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
        if (nk < (size_t)q.dM || nk < (size_t)q.dD) { rc = fail(ctx, AEFFT_EINVAL, "aefft_net_create: degenerate kernel"); break; }
        q.goff = goff; goff += 2 * nk + q.dM + q.dD;
        // c|f, Dc|Df, C|F and dc|df are each ONE allocation so that both kernels of a pair go through one launch
        if ((rc = net_alloc_t(n, &q.c, 2 * nk)) || (rc = net_alloc_t(n, &q.Dc, 2 * nk))) break;
        q.f = q.c + nk; q.Df = q.Dc + nk;
        if ((rc = net_alloc_t(n, &q.b, q.dM)) || (rc = net_alloc_t(n, &q.Db, q.dM)) || (rc = net_alloc_t(n, &q.p, q.dD)) || (rc = net_alloc_t(n, &q.Dp, q.dD))) break;
        const size_t W = (size_t)q.dM * q.dD * q.P;
        if ((rc = net_alloc_t(n, &q.C, 2 * W))) break;
        q.F = q.C + W;
        q.spectra_valid = false;
        const size_t BDP = (size_t)n->Bc * q.dD * q.P, BMP = (size_t)n->Bc * q.dM * q.P;
        if (q.s == 1 && l > 0) q.X = n->pr[l - 1].H;
        else if ((rc = net_alloc_t(n, &q.X, BDP))) break;
        if ((rc = net_alloc_t(n, &q.H, BMP)) || (rc = net_alloc_t(n, &q.O, BDP))) break;
        q.Oc = nullptr;
        if ((rc = net_alloc_t(n, &q.S, (size_t)q.dD * q.dD * q.P)) || (rc = net_alloc_t(n, &q.G, (size_t)q.dD * q.dD * q.P)) || (rc = net_alloc_t(n, &q.dc, 2 * W))) break;
        q.df = q.dc + W;
        q.part = nullptr;
        if (pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny)) { if ((rc = net_alloc_t(n, &q.part, kgrad_partial_floats(2L * q.dM * q.dD, q.Nx, q.Ny, q.Nk, q.Nl)))) break; }
        else n->pruned = false;
        maxS = std::max(maxS, (size_t)q.dD * q.dD * q.P); maxBDP = std::max(maxBDP, BDP); maxW = std::max(maxW, W);
        esoff[l] = soff; soff += 2 * (size_t)q.dD;
        maxReal = std::max(maxReal, (size_t)q.dM * q.dD * q.Nx * q.Ny);
        maxMid = std::max(maxMid, (size_t)q.dM * q.dD * q.Nx * (q.Ny / 2));
        maxDen = std::max(maxDen, gradient_diff_ws_floats(q.dM, q.dD, q.Nk, q.Nl));
        maxSmall = std::max(maxSmall, 2 * nk + q.dM + q.dD + 64);
        dD = q.dM; nx = q.Nx; ny = q.Ny;
    }
    if (rc == AEFFT_OK) {
        maxMid = std::max(maxMid, (size_t)n->B * n->D * n->Nx * (n->Ny / 2));
        void* dummy;
        // size the context workspaces once so nothing reallocates inside a step
        if ((rc = ws_get(ctx, WS_MID, sizeof(float2) * maxMid, &dummy)) == AEFFT_OK &&
            (rc = ws_get(ctx, WS_MID3, sizeof(float2) * (size_t)n->B * n->D * n->Nx * (n->Ny / 2), &dummy)) == AEFFT_OK &&
            (rc = ws_get(ctx, WS_DEN, sizeof(float) * maxDen, &dummy)) == AEFFT_OK &&
            (rc = ws_get(ctx, WS_SMALL, sizeof(float) * maxSmall, &dummy)) == AEFFT_OK &&
            (rc = net_alloc_t(n, &n->real, n->pruned ? 64 : maxReal)) == AEFFT_OK &&
            (rc = net_alloc_t(n, &n->mse_slots, (size_t)n->L * MSE_SLOTS * MSE_SLOT_STRIDE)) == AEFFT_OK &&
            (rc = net_alloc_t(n, &n->grad, goff + 2 * (size_t)n->L)) == AEFFT_OK && (rc = net_alloc_t(n, &n->scratch, soff)) == AEFFT_OK) {
            n->scratch_n = soff; n->mse_pre = n->scratch; n->mse_post = n->scratch + n->L;
            for (int l = 0; l < n->L; ++l) n->pr[l].es = n->scratch + esoff[l];
            n->grad_n = goff;
            n->Xf = n->pr[0].X;
            if (n->D <= OPC - 1) {
                const Pair& q0 = n->pr[0];
                if (rc == AEFFT_OK) rc = net_alloc_t(n, &n->A0hat, (size_t)OPC * q0.dD * q0.P);
                if (rc == AEFFT_OK) rc = net_alloc_t(n, &n->Mhat, (size_t)OPC * OPC * q0.P);
                if (rc == AEFFT_OK && launch_basis_fill(n->A0hat, q0.dD, q0.P, ctx->stream) != hipSuccess) rc = fail(ctx, AEFFT_EHIP, "basis_fill");
                if (rc == AEFFT_OK) rc = build_chain_items(n);
            }
            // compact decoder outputs (training step): the coarsest pair's grid
            const Pair& qc = n->pr[n->L - 1];
            n->NxC = qc.Nx; n->NyC = qc.Ny; n->Pc = qc.P;
            for (int l = 0; l < n->L && rc == AEFFT_OK; ++l) {
                Pair& q = n->pr[l];
                if (rc == AEFFT_OK) rc = net_alloc_t(n, &q.beta, (size_t)q.dD);
                if (q.P == n->Pc) q.Oc = q.O;            // already on the coarsest grid: nothing to compact
                else rc = net_alloc_t(n, &q.Oc, (size_t)n->Bc * q.dD * n->Pc);
                if (rc == AEFFT_OK && q.Nk == q.Nl && (q.Nk == 3 || q.Nk == 5)) {
                    const size_t tt = (size_t)(2 * q.Nk - 1) * (2 * q.Nk - 1);
                    q.Qn = kgrad_group_chunks((long)q.dD * q.dD, q.Nx, q.Ny);       // room for the row chunks' partial sums
                    rc = net_alloc_t(n, &q.Q, (size_t)q.dD * q.dD * tt * q.Qn);
                }
            }
        }
    }
    if (rc != AEFFT_OK) { aefft_net_destroy(n); return rc; }
    hipError_t e = hipMemsetAsync(n->mse_slots, 0, sizeof(float) * n->L * MSE_SLOTS * MSE_SLOT_STRIDE, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(n->grad + n->grad_n, 0, sizeof(float) * 2 * n->L, ctx->stream);      // (the MSE tail of the packed buffer: zero before the first step)
    if (e != hipSuccess) { aefft_net_destroy(n); return fail(ctx, AEFFT_EHIP, "memset(mse slots)", e); }
    for (auto& q : n->pr) {
        const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
        e = hipMemsetAsync(q.c, 0, nk * 4, ctx->stream); if (e) break;
        e = hipMemsetAsync(q.f, 0, nk * 4, ctx->stream); if (e) break;
        e = hipMemsetAsync(q.b, 0, q.dM * 4, ctx->stream); if (e) break;
        e = hipMemsetAsync(q.p, 0, q.dD * 4, ctx->stream); if (e) break;
    }
    if (e != hipSuccess) { aefft_net_destroy(n); return fail(ctx, AEFFT_EHIP, "memset weights", e); }
    if (ensure_aux(ctx) != AEFFT_OK) { aefft_net_destroy(n); return AEFFT_EHIP; }
    *out = n;
    return aefft_net_reset_momentum(n);
}

extern "C" int aefft_net_npairs(aefft_net* n) { return n ? n->L : -1; }
extern "C" int aefft_net_pair_shape(aefft_net* n, int l, int* dD, int* dM, int* Nk, int* Nl)
{
    if (!n || l < 0 || l >= n->L) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_pair_shape: bad pair index");
    const Pair& q = n->pr[l];
    if (dD) *dD = q.dD;
    if (dM) *dM = q.dM;
    if (Nk) *Nk = q.Nk;
    if (Nl) *Nl = q.Nl;
    return AEFFT_OK;
}

extern "C" int aefft_net_reset_momentum(aefft_net* n)
{
    if (!n) return AEFFT_EINVAL;
    n->upd_after_fwd = false;          // (w + D no longer is the previous weight)
    aefft_ctx* ctx = n->ctx;
    for (auto& q : n->pr) {
        const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
        HIPCHK(ctx, hipMemsetAsync(q.Dc, 0, nk * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(q.Df, 0, nk * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(q.Db, 0, q.dM * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(q.Dp, 0, q.dD * 4, ctx->stream));
    }
    return AEFFT_OK;
}

extern "C" int aefft_net_set_pair(aefft_net* n, int l, const float* c_h, const float* b_h, const float* f_h, const float* p_h)
{
    if (!n || l < 0 || l >= n->L || !c_h || !b_h || !f_h || !p_h) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_set_pair: bad argument");
    n->upd_after_fwd = false;
    aefft_ctx* ctx = n->ctx;
    Pair& q = n->pr[l];
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    HIPCHK(ctx, hipMemcpyAsync(q.c, c_h, nk * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.f, f_h, nk * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.b, b_h, q.dM * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.p, p_h, q.dD * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // host buffers may be pageable / reused by the caller
    q.spectra_valid = false; q.G_valid = false; n->packed_valid = false; n->chain_valid = false;
    return AEFFT_OK;
}

extern "C" int aefft_net_get_pair(aefft_net* n, int l, float* c_h, float* b_h, float* f_h, float* p_h)
{
    if (!n || l < 0 || l >= n->L) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_get_pair: bad argument");
    aefft_ctx* ctx = n->ctx;
    Pair& q = n->pr[l];
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    if (c_h) HIPCHK(ctx, hipMemcpyAsync(c_h, q.c, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (f_h) HIPCHK(ctx, hipMemcpyAsync(f_h, q.f, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (b_h) HIPCHK(ctx, hipMemcpyAsync(b_h, q.b, q.dM * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (p_h) HIPCHK(ctx, hipMemcpyAsync(p_h, q.p, q.dD * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AEFFT_OK;
}

// kernels -> spectra for both tensors of a pair (StoreLoad_cfreq first pass, fft_backproplib.cu:1150-1152; :1274-1282)
static int pair_spectra(aefft_net* n, Pair& q)
{
    const long planes = (long)q.dM * q.dD;
    if (pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny)) return do_pad_r2c(n->ctx, q.c, q.C, nullptr, 2 * planes, q.Nx, q.Ny, q.Nk, q.Nl);
    RET_IF(do_pad_r2c(n->ctx, q.c, q.C, n->real, planes, q.Nx, q.Ny, q.Nk, q.Nl));
    return do_pad_r2c(n->ctx, q.f, q.F, n->real, planes, q.Nx, q.Ny, q.Nk, q.Nl);
}

static int ensure_spectra(aefft_net* n, Pair& q)
{
    if (q.spectra_valid) return AEFFT_OK;
    RET_IF(pair_spectra(n, q));
    q.spectra_valid = true;
    return AEFFT_OK;
}

extern "C" int aefft_net_pair_spectra(aefft_net* n, int l, float** C_d, float** F_d)
{
    if (!n || l < 0 || l >= n->L) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_pair_spectra: bad argument");
    RET_IF(ensure_spectra(n, n->pr[l]));
    if (C_d) *C_d = reinterpret_cast<float*>(n->pr[l].C);
    if (F_d) *F_d = reinterpret_cast<float*>(n->pr[l].F);
    return AEFFT_OK;
}

extern "C" int aefft_net_store_spectra(aefft_net* n, int l, float* C_h, float* F_h)
{
    if (!n || l < 0 || l >= n->L) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_store_spectra: bad argument");
    aefft_ctx* ctx = n->ctx;
    Pair& q = n->pr[l];
    RET_IF(ensure_spectra(n, q));
    const size_t W = (size_t)q.dM * q.dD * q.P * sizeof(float2);
    if (C_h) HIPCHK(ctx, hipMemcpyAsync(C_h, q.C, W, hipMemcpyDeviceToHost, ctx->stream));
    if (F_h) HIPCHK(ctx, hipMemcpyAsync(F_h, q.F, W, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AEFFT_OK;
}

// load_cfreq semantics (fft_backproplib.cu:1131-1141): the cached SPECTRA are the source of truth;
// the coordinate-space kernels are re-derived from them (export_cfreq, :1166) to stay consistent.
extern "C" int aefft_net_load_spectra(aefft_net* n, int l, const float* C_h, const float* b_h, const float* F_h, const float* p_h)
{
    if (!n || l < 0 || l >= n->L || !C_h || !b_h || !F_h || !p_h) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_load_spectra: bad argument");
    n->upd_after_fwd = false;
    aefft_ctx* ctx = n->ctx;
    Pair& q = n->pr[l];
    const size_t W = (size_t)q.dM * q.dD * q.P * sizeof(float2);
    HIPCHK(ctx, hipMemcpyAsync(q.C, C_h, W, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.F, F_h, W, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.b, b_h, q.dM * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q.p, p_h, q.dD * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    q.spectra_valid = true; q.G_valid = false; n->packed_valid = false; n->chain_valid = false;
    RET_IF(aefft_kernel_export(ctx, reinterpret_cast<const float*>(q.C), q.c, q.dM, q.dD, q.Nk, q.Nl, q.Nx, q.Ny));
    RET_IF(aefft_kernel_export(ctx, reinterpret_cast<const float*>(q.F), q.f, q.dD, q.dM, q.Nk, q.Nl, q.Nx, q.Ny));
    return AEFFT_OK;
}

static int mse_flush(aefft_net* n);
static int mark_step_point(aefft_net* n);
// lazy: encoder outputs that are only consumed through pool_fft are computed on the pooled grid alone (the bins the crop
// discards are never formed; aefft_net_get_layer recomputes such a layer on demand).  The training step uses it.
// chain_kernel (opform_kernels.hip) runs the network on the basis frames in one launch; its coarsest-grid workgroups read a
// bin-major copy of the kernel spectra (kspec_packed_kernel).  Served when the coarsest grid is small (large ones stream better
// layer by layer) and the channel counts fit the kernel's LDS tiles.
static int build_chain_items(aefft_net* n)
{
    const int L = n->L;
    bool dims_ok = true;
    for (const Pair& q : n->pr) dims_ok = dims_ok && q.dD <= 128 && q.dM <= 128;
    for (int l = 0; l + 1 < L; ++l) dims_ok = dims_ok && 2 * n->pr[l].dM * (int)OPC * 8 <= 6144;
    if (!dims_ok || n->pr[L - 1].P > 16384) return AEFFT_OK;
    // the bin-major copy for the coarsest-grid items: C_0 .. C_{L-1}, F_{L-1} .. F_0 in chain order, segments padded to even sizes
    bool pk = n->pr[0].Nk == n->pr[0].Nl && (n->pr[0].Nk == 3 || n->pr[0].Nk == 5) && 2 * L <= 16;
    for (const Pair& q : n->pr) pk = pk && q.Nk == n->pr[0].Nk && q.Nl == n->pr[0].Nk && (((q.dM * q.dD + 1) & ~1) <= CH_VMAX * CH_VMAX);      // (matrices are read from the record in place: only the row counts are bounded)
    if (pk) {
        PackArgs& pa = n->pack;
        int off = 0, ns = 0;
        // (with each tensor the gradient and momentum of the same taps: a fused update reads the taps through them, TapUpd)
        for (int l = 0; l < L; ++l) { const Pair& q = n->pr[l]; pa.seg[ns++] = PackSeg{q.c, q.dM * q.dD, l, off, n->grad + q.goff, q.Dc}; off += (q.dM * q.dD + 1) & ~1; }
        for (int l = L - 1; l >= 0; --l) {
            const Pair& q = n->pr[l];
            pa.seg[ns++] = PackSeg{q.f, q.dM * q.dD, l, off, n->grad + q.goff + (size_t)q.dM * q.dD * q.Nk * q.Nl, q.Df};
            off += (q.dM * q.dD + 1) & ~1;
        }
        pa.nseg = ns; pa.L = L; pa.E = off; pa.Nk = n->pr[0].Nk;
        for (int l = 0; l < L; ++l) { pa.Nx[l] = n->pr[l].Nx; pa.Ny[l] = n->pr[l].Ny; }
        pa.NxC = n->pr[L - 1].Nx; pa.NyC = n->pr[L - 1].Ny; pa.Pc = n->pr[L - 1].P; pa.tw = n->ctx->tw;
        if ((size_t)pa.Pc * pa.E * sizeof(float2) <= (size_t)1 << 30) {
            RET_IF(net_alloc_t(n, &n->Wp, (size_t)pa.Pc * pa.E));
            pa.Wp = n->Wp;
            if (2 * (L - 1) <= 8)
                for (int l = 0; l + 1 < L; ++l) RET_IF(net_alloc_t(n, &n->pr[l].Cc, (size_t)n->pr[l].dM * n->pr[l].dD * n->pr[l + 1].P));
            for (int l = 0; l < L; ++l)
                for (int k = 0; k < 2; ++k) {
                    if (l > 0) RET_IF(net_alloc_t(n, &n->pr[l].opA[k], (size_t)OPC * n->pr[l].dD * n->pr[l].P));
                    RET_IF(net_alloc_t(n, &n->pr[l].opO[k], (size_t)OPC * n->pr[l].dD * pa.Pc));
                }
        }
    }
    return AEFFT_OK;
}

// problems of the spectra launch that serve the operator chain: Cc_l for l < L-1 (C sampled where the next pair's grid lands)
static int cc_problems(aefft_net* n, PrunedGroup& pg, int first, double* bytes)
{
    int k = first;
    for (int l = 0; l + 1 < n->L; ++l) {
        Pair& q = n->pr[l];
        const Pair& nx = n->pr[l + 1];
        pg.q[k] = PrunedProb{q.c, q.Cc, (long)q.dM * q.dD, nx.Nx, nx.Ny, 1.0f, q.Nx, q.Ny};
        if (bytes) *bytes += (double)q.dM * q.dD * (nx.P * 8.0 + q.Nk * q.Nl * 4.0);
        ++k;
    }
    return k;
}

// G'_l = F_l.C_l / (dM dD) [dD][dD][P] of the STORED weights of every pair, into Pair::G: the spectrum of the (2Nk-1)^2-tap kernel f (*) c, taps
// formed inside the transforming workgroups (gspec_gbody).  false: shapes not served.
static int gprime_from_taps(aefft_net* n, bool* done)
{
    *done = false;
    aefft_ctx* ctx = n->ctx;
    if (n->L > 8 || n->pr[0].Nk != n->pr[0].Nl || (n->pr[0].Nk != 3 && n->pr[0].Nk != 5)) return AEFFT_OK;
    PrunedGroup pg{};
    GtapsGroup tg{};
    double bytes = 0;
    const int TT = (2 * n->pr[0].Nk - 1) * (2 * n->pr[0].Nk - 1);
    bool chunked = false;                                  // some plane is transformed by several row-chunk workgroups: the taps are formed once, in a launch in front
    for (int l = 0; l < n->L; ++l) {
        Pair& q = n->pr[l];
        if (q.Nk != n->pr[0].Nk || q.Nl != n->pr[0].Nk || !pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny) || (double)q.dD * q.dD * q.P * 8.0 >= 4294967296.0) return AEFFT_OK;
        pg.q[l] = PrunedProb{nullptr, q.G, (long)q.dD * q.dD, q.Nx, q.Ny, 1.0f, 0, 0};
        pg.gsrc[l] = GtapSrc{q.c, q.f, q.dM, q.dD, 1.0f / ((float)q.dM * (float)q.dD)};
        bytes += (double)q.dD * q.dD * q.P * 8.0 + 2.0 * q.dM * q.dD * q.Nk * q.Nl * 4.0;
        chunked = chunked || q.Nx > 64;
    }
    if (chunked) {
        // taps once per plane, then their spectra as an ordinary pruned transform of (2Nk-1)^2-tap kernels (its own kernel instantiation:
        // the planar-spectra launch keeps its code)
        if (!n->gtaps) {
            size_t nt = 0;
            for (const Pair& q : n->pr) nt += (size_t)q.dD * q.dD * TT;
            RET_IF(net_alloc_t(n, &n->gtaps, nt));
        }
        PrunedGroup pt{};
        float* o = n->gtaps;
        for (int l = 0; l < n->L; ++l) {
            const Pair& q = n->pr[l];
            tg.gs[l] = pg.gsrc[l]; tg.out[l] = o;
            pt.q[l] = PrunedProb{o, q.G, (long)q.dD * q.dD, q.Nx, q.Ny, 1.0f, 0, 0};
            o += (size_t)q.dD * q.dD * TT;
        }
        tg.n = pt.n = n->L;
        hipError_t e;
        {
            Bracket br(ctx, KID_KSPEC, bytes);
            e = launch_gtaps_group(tg, n->pr[0].Nk, ctx->cur);
            if (e == hipSuccess) e = launch_kspec_group_taps(pt, ctx->tw, 2 * n->pr[0].Nk - 1, ctx->cur);
        }
        if (e == hipSuccess) { *done = true; return AEFFT_OK; }
        if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "G'(stored taps)", e);
        (void)hipGetLastError();                           // (shapes these launches do not serve: taps formed in the transforming workgroups, below)
    }
    pg.n = n->L;
    hipError_t e;
    {
        Bracket br(ctx, KID_KSPEC, bytes);
        e = launch_kspec_group(pg, ctx->tw, n->pr[0].Nk, n->pr[0].Nl, ctx->cur, nullptr, nullptr);
    }
    if (e == hipSuccess) { *done = true; return AEFFT_OK; }
    if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "G'(taps)", e);
    (void)hipGetLastError();
    return AEFFT_OK;
}

// the bin-major record Wp (and the compact Cc planes) of the CURRENT weights
static int ensure_packed(aefft_net* n)
{
    if (!n->Wp || n->packed_valid) return AEFFT_OK;
    aefft_ctx* ctx = n->ctx;
    double bytes = (double)n->pack.Pc * n->pack.E * 8.0;
    hipError_t e;
    if (n->pr[0].Cc) {
        PrunedGroup pg{};
        pg.n = cc_problems(n, pg, 0, &bytes);
        n->pack.upd = 0;
        Bracket br(ctx, KID_KSPEC, bytes);
        e = launch_kspec_group(pg, ctx->tw, n->pr[0].Nk, n->pr[0].Nl, ctx->cur, &n->pack, nullptr);
    } else {
        Bracket br(ctx, KID_KSPEC, bytes);
        e = launch_kspec_packed(n->pack, ctx->cur);
    }
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "kspec_packed", e);
    n->packed_valid = true;
    return AEFFT_OK;
}

// the training step's forward runs as ONE chain launch on the basis frames (chain_kernel) under these switches
static bool chain_switches_ok() { return !(dev_flags & (AEFFT_F_NOCHAIN | AEFFT_F_NOLAZY | AEFFT_F_NOCOMPACT | AEFFT_F_NOGROUP | AEFFT_F_NOMFMA | AEFFT_F_NOFUSECROP)); }

// The training step runs in operator form (opform_kernels.hip) when every pair has the Q-path gradient (equal square 3x3 / 5x5
// supports with pruned transforms) and the input has at most OPC-1 channels.
static bool op_eligible(const aefft_net* n)
{
    if (flag(AEFFT_F_NOOPFORM) || flag(AEFFT_F_NOQPATH) || !n->A0hat || n->L > 8) return false;
    const Pair& q0 = n->pr[0];
    if (q0.Nk != q0.Nl || (q0.Nk != 3 && q0.Nk != 5)) return false;
    for (const Pair& q : n->pr) if (q.Nk != q0.Nk || q.Nl != q0.Nl || !q.Q || !pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny)) return false;
    // channel counts the operator-form kernels' LDS tiles take (msgrad_kernel: 2*OPC*dD*8 complex; opmse: OPC*(dD+dM)*4 complex): a
    // launch declined in the middle of step_apply would leave a fused update half applied, so the step form is decided here
    for (const Pair& q : n->pr) if (q.dD > 256 || q.dM > 512 || q.dD + q.dM > 1024) return false;
    return true;
}
static bool chain_switches_ok();

// operator form: where pair l's operators of the step in progress are (A_l [OPC][dD][P], O^_l [OPC][dD][PO] on the grid nxo x nyo)
struct OpView { const float2 *A, *O; int nxo, nyo; long PO; };
static bool op_mode(const aefft_net* n) { return n->op_state || n->op_chain; }
static OpView op_view(const aefft_net* n, int l)
{
    const Pair& q = n->pr[l];
    if (n->op_chain) return OpView{l == 0 ? n->A0hat : q.opA[n->op_fwd], q.opO[n->op_fwd], n->NxC, n->NyC, n->Pc};
    const bool st = q.O_stale;
    return OpView{q.X, st ? q.Oc : q.O, st ? n->NxC : q.Nx, st ? n->NyC : q.Ny, st ? n->Pc : q.P};
}
static void fill_chain(aefft_net* n, ChainArgs& ca, int set, double* bytes)
{
    const int L = n->L;
    const bool cc = L == 1 || n->pr[0].Cc != nullptr;
    for (int l = 0; l < L; ++l) {
        Pair& q = n->pr[l];
        ca.lv[l] = ChainLevel{q.C, q.F, q.b, q.p, l == 0 ? n->A0hat : q.opA[set], q.opO[set], q.dD, q.dM, q.Nx, q.Ny, q.P, cc ? q.Cc : nullptr};
        const double cb = (l + 1 < L) ? (double)n->pr[l + 1].P : (double)q.P;
        if (bytes) *bytes += ((double)q.dM * q.dD * (cb + n->Pc) + (double)OPC * q.dD * (q.P + n->Pc)) * 8.0;
    }
    ca.L = L; ca.D0 = n->D; ca.Pc = n->Pc; ca.Wp = n->Wp; ca.E = n->pack.E;
}

// the reconstruction's inverse FFT (fft_backproplib.cu:1373) on ctx->cur; operator form: the per-frame spectra are expanded first
static int launch_recon(aefft_net* n, float* recon_d, int wsid)
{
    aefft_ctx* ctx = n->ctx;
    Pair& q = n->pr[0];
    const OpView ov = op_mode(n) ? op_view(n, 0) : OpView{nullptr, q.O_stale ? q.Oc : q.O, q.O_stale ? n->NxC : q.Nx, q.O_stale ? n->NyC : q.Ny, 0};
    const float2* src = ov.O;
    const int nxo = ov.nxo, nyo = ov.nyo;
    if (op_mode(n)) {
        static_assert(OPIN_COLS == OPC, "operator width");
        const long PO = bins(nxo, nyo);
        if ((double)n->B * q.dD * PO * 8.0 > 16e6) {
            // large supports (no pooling: the decoder output lives on the whole grid): the per-frame spectra O_0,b = O^_0 [x_b; 1] are
            // written out once by a coalesced pass (7 plane-ordered loads per output) and the inverse transform reads them back.  Evaluated
            // inside the column pass instead, the same 7 loads are strided 128-byte pieces: 1.1 ms against 0.2 ms at cfg3-P1.
            if (!n->recon_exp) RET_IF(net_alloc_t(n, &n->recon_exp, (size_t)n->B * q.dD * PO));
            {
                Bracket br(ctx, KID_OPFORM, ((double)OPC * q.dD * PO + (double)n->B * q.dD * PO + (double)n->B * q.dD * q.P) * 8.0);
                hipError_t e = launch_recon_expand(src, n->Xf, n->recon_exp, n->B, q.dD, q.Nx, q.Ny, nxo, nyo, ctx->cur);
                if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "recon_expand", e);
            }
            return do_c2r(ctx, n->recon_exp, recon_d, (long)n->B * q.dD, nxo, nyo, n->Nx, n->Ny, 1.0f / ((float)n->Nx * (float)n->Ny), wsid);
        }
        // small supports: O_0,b = O^_0 [x_b; 1] is evaluated inside the column pass of the inverse transform (no stored planes)
        const OpIn op{src, n->Xf, q.dD, q.Nx, q.Ny};
        return do_c2r(ctx, nullptr, recon_d, (long)n->B * q.dD, nxo, nyo, n->Nx, n->Ny, 1.0f / ((float)n->Nx * (float)n->Ny), wsid, &op);
    }
    return do_c2r(ctx, src, recon_d, (long)n->B * q.dD, nxo, nyo, n->Nx, n->Ny, 1.0f / ((float)n->Nx * (float)n->Ny), wsid);
}

// op: run the network on the OPC basis frames (the activation buffers then hold the per-bin operators A_l, O^_l) -- the
// frames themselves only go through the input transform, the second moments and the reconstruction.
static int net_forward(aefft_net* n, const float* frames_d, float* recon_d, bool lazy, bool op = false)
{
    if (!n || !frames_d) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_forward: bad argument");
    aefft_ctx* ctx = n->ctx;
    const int BF = n->B;                       // frames
    const int B = op ? (int)OPC : n->B;        // columns of the activation buffers
    const int L = n->L;
    struct BiasColGuard { aefft_ctx* c; ~BiasColGuard() { c->biasColP1 = 0; } } guard{ctx};
    ctx->biasColP1 = op ? (int)OPC : 0;        // conv_k biases: the affine column only
    RET_IF(join_recon(ctx));
    n->xx_done = false; n->ox_done = 0;
    n->upd_after_fwd = false;
    // the whole network on the basis frames in one launch (chain_kernel): hidden layers not materialised, decoder outputs on the
    // coarsest grid's support, operators in their own buffers
    const bool chain_plan = op && lazy && n->Wp && (n->compact || L == 1) && chain_switches_ok();
    // (the chain launch reads the bin-major record Wp and the compact Cc planes only: planar spectra that a training step in operator
    // form does not refresh are formed when something else asks for them)
    const bool chain_cc = chain_plan && (L == 1 || n->pr[0].Cc != nullptr);
    n->op_state = op && !chain_plan;
    n->op_chain = chain_plan;
    n->act_stale = chain_plan;
    for (int l = 0; l < L; ++l) if (!(chain_cc || (chain_plan && l == L - 1 && L > 1))) RET_IF(ensure_spectra(n, n->pr[l]));
    // the reconstruction's side stream forks behind the last launch in front of the gradient kernels -- the input transform's column
    // pass, or the chain launch when the operators of the current weights are not at hand (first step, weights set from outside) --
    // through that dispatch's own completion signal
    // (reconstructions beyond ~256 MB -- 32 frames of 1024^2 -- stay on the context stream: beside their row pass the pruned inverse transform
    // of S stretches from 42 to 145 us and the side stream costs more than it hides, 1.084 vs 1.057 ms per cfg5 step; at cfg3 it saves 15 of 203 us)
    // ... and reconstructions below ~8 MB (cfg2: one 256^2 frame, 11 us of kernels) stay there as well: the fork and join packets cost more than the
    // two kernels they would hide (0.074 vs 0.076 ms per cfg2 step)
    const double recon_bytes = (double)n->B * n->D * n->Nx * n->Ny * 4.0;
    const bool overlap_pays = recon_bytes <= 256e6 && (recon_bytes >= 8e6 || flag(AEFFT_F_SMALLOVERLAP));
    const bool want_fork = chain_plan && recon_d && ctx->aux[0] != nullptr && !flag(AEFFT_F_NOOVERLAP) && overlap_pays && !ctx->prof &&
                           !(n->input_ready && !flag(AEFFT_F_NODEFER)) && ctx->cur == ctx->stream;
    const bool need_chain = chain_plan && !n->chain_valid;
    bool fork_recorded = false;
    // encoder (fft_backproplib.cu:1340-1357): R2C fused with pair 0's pooling, then pool -> conv per pair
    const bool prefetch = lazy && n->input_ready && n->X0alt && ctx->aux[1] != nullptr && !ctx->prof && !flag(AEFFT_F_NOPREFETCH);
    if (prefetch) {
        // The caller guarantees the frames are complete: their R2C goes to a side stream and may overlap the tail of the previous
        // step.  It writes the OTHER input-spectra buffer (the current one is still read by that tail), which was last read two
        // steps ago: wait for that step's end only.
        std::swap(n->Xf, n->X0alt);
        if (n->ev_end_valid[n->step_no & 1]) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[1], n->ev_end[n->step_no & 1], 0));
        // not earlier than the end of the previous step's gradient half: that is where a data-parallel run waits for its
        // all-reduce (an otherwise idle gap), and what follows on this stream (update, spectra, MSE) is latency-bound
        if (n->ev_mid_valid) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[1], n->ev_mid, 0));
        ctx->cur = ctx->aux[1];
        const int rc = do_r2c(ctx, frames_d, n->Xf, (long)BF * n->D, n->Nx, n->Ny, n->pr[0].Nx, n->pr[0].Ny, WS_MID2);
        ctx->cur = ctx->stream;
        RET_IF(rc);
        HIPCHK(ctx, hipEventRecord(n->ev_r2c, ctx->aux[1]));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, n->ev_r2c, 0));
    } else {
        const bool fork_r2c = want_fork && !need_chain;
        RET_IF(do_r2c(ctx, frames_d, n->Xf, (long)BF * n->D, n->Nx, n->Ny, n->pr[0].Nx, n->pr[0].Ny, WS_MID, fork_r2c ? ctx->ev_fork : nullptr));
        fork_recorded = fork_r2c;
    }
    n->pr[0].X = n->op_state ? n->A0hat : n->Xf;
    bool chained = false;
    if (chain_plan) {
        if (need_chain) {
            RET_IF(ensure_packed(n));
            ChainArgs ca{};
            double bytes = 0;
            fill_chain(n, ca, n->op_set, &bytes);
            Bracket br(ctx, KID_CHAIN, bytes);
            const bool fork_here = want_fork && !fork_recorded;
            hipError_t e = launch_chain(ca, ctx->cur, fork_here ? ctx->ev_fork : nullptr);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "chain", e);
            fork_recorded = fork_recorded || fork_here;
            n->chain_valid = true;
        }
        n->op_fwd = n->op_set;
        for (int l = 0; l < L; ++l) { Pair& q = n->pr[l]; q.H_stale = true; q.O_stale = q.P != n->Pc; }      // (what ensure_frames leaves in the activation buffers)
        chained = true;
    }
    for (int l = 0; l < L && !chained; ++l) {
        Pair& q = n->pr[l];
        // the next pair's spectral down-sampling (pool_fft, :1346) is written by this conv's epilogue: no resize launch
        const bool fuse = (l + 1 < L) && n->pr[l + 1].s != 1 && n->fuse_crop && !flag(AEFFT_F_NOFUSECROP);
        q.H_stale = false;
        if (lazy && !op && l == L - 1 && q.G_valid && !flag(AEFFT_F_NOGFWD)) {
            // innermost pair of a training step: its hidden layer feeds only its own decoder conv, and the previous step left
            // the collapsed operator of the CURRENT weights behind (G = F.C/(dM dD) in S, DC bias in beta): O = G X + beta below,
            // a quarter of the arithmetic and bytes of conv_k o conv_k, no H.
            q.H_stale = true;
            continue;
        }
        if (fuse && lazy) {
            const bool nolazy = flag(AEFFT_F_NOLAZY);
            const Pair& nx = n->pr[l + 1];
            bool done = false;
            if (!nolazy) RET_IF(do_conv_pooled(ctx, q.X, q.C, q.b, nx.X, B, q.dM, q.dD, q.Nx, q.Ny, nx.Nx, nx.Ny, &done));
            if (done) { q.H_stale = true; continue; }
        }
        if (fuse) { const Pair& nx = n->pr[l + 1]; RET_IF(do_conv(ctx, q.X, q.C, q.b, q.H, B, q.dM, q.dD, q.Nx, q.Ny, nx.X, nx.Nx, nx.Ny)); }
        else RET_IF(do_conv(ctx, q.X, q.C, q.b, q.H, B, q.dM, q.dD, q.Nx, q.Ny));
        if (!fuse && l + 1 < L && n->pr[l + 1].s != 1) {
            const Pair& nx = n->pr[l + 1];
            RET_IF(do_resize(ctx, q.H, nx.X, (long)B * nx.dD, nx.Nxin, nx.Nyin, nx.Nx, nx.Ny));
        }
    }
    // decoder (:1356-1361): conv then zero-pad up-sampling.  The up-sampled tensor is never stored: the next
    // decoder conv (and the final C2R) read the small spectrum through the zero-pad index map.
    const bool nocompact = flag(AEFFT_F_NOCOMPACT);
    bool compact = lazy && n->compact && !nocompact && L > 1;
    for (int l = L - 1; l >= 0 && !chained; --l) {
        Pair& q = n->pr[l];
        q.O_stale = false;
        if (l == L - 1) {
            if (q.H_stale) {                       // (set above: G route)
                Contract k{};
                k.A = q.G; k.a_r = (long)q.dD * q.P; k.a_k = q.P;
                k.B = q.X; k.b_k = q.P; k.b_c = (long)q.dD * q.P;
                k.Out = q.O; k.o_r = q.P; k.o_c = (long)q.dD * q.P;
                k.R = q.dD; k.C = B; k.K = q.dD; k.P = q.P;
                k.bias = q.beta; k.biasScale = (float)q.Nx * (float)q.Ny; k.biasAfterFirst = true;
                // the batch-first gradient term S = -sum_b X X^H needs only the encoder outputs: it shares this launch
                // (the decoder chain that follows is a sequence of small dependent launches)
                n->xx_done = false; n->ox_done = 0;
                if (compact && n->pr[0].P != n->Pc && L + 1 <= 8 && !flag(AEFFT_F_NOGROUP)) {
                    Contract qs[8];
                    qs[0] = k;
                    for (int l2 = 0; l2 < L; ++l2) { Pair& q2 = n->pr[l2]; qs[1 + l2] = mk_XXneg(q2.X, q2.S, B, q2.dD, q2.P); }
                    RET_IF(do_contract_group(ctx, qs, L + 1, L + 1, 0));
                    n->xx_done = true;
                } else
                RET_IF(do_contract(ctx, k));
            } else RET_IF(do_conv(ctx, q.H, q.F, q.p, q.O, B, q.dD, q.dM, q.Nx, q.Ny));
            continue;
        }
        const Pair& in = n->pr[l + 1];
        if (compact && q.P != n->Pc) {
            // Up-sampled spectra are zero outside the image of the coarsest grid, and conv_k maps zero to zero (the bias sits on
            // the DC bin, inside it): every decoder output lives on those Pc bins.  The training step computes and stores only them:
            //   Oc_l[b][d][s] = sum_m F_l[d][m][map_l(s)] * Oc_{l+1}[b][m][s] / dD + p[d] Nx Ny [s == 0]
            Contract k{};
            k.A = q.F; k.a_r = (long)q.dM * q.P; k.a_k = q.P;
            k.B = in.Oc; k.b_k = n->Pc; k.b_c = (long)q.dM * n->Pc;
            k.Out = q.Oc; k.o_r = n->Pc; k.o_c = (long)q.dD * n->Pc;
            k.R = q.dD; k.C = B; k.K = q.dM; k.P = n->Pc;
            k.preDivB = (float)q.dD;
            k.bias = q.p; k.biasScale = (float)q.Nx * (float)q.Ny; k.biasAfterFirst = true;
            k.gdNx = q.Nx; k.gdNy = q.Ny; k.gdNxs = n->NxC; k.gdNys = n->NyC; k.gdMask = 1;
            if (n->xx_done && !op && !flag(AEFFT_F_NOGROUP) && !flag(AEFFT_F_NOMFMA)) {
                // S = -sum_b X X^H is already out: the support term of the NEXT-inner pair (its decoder output is final) rides along
                Pair& qi = n->pr[l + 1];
                Contract qs[2] = {k, qi.O_stale ? mk_OX(qi.Oc, qi.X, qi.S, B, qi.dD, qi.P, n->Pc, qi.Nx, qi.Ny, n->NxC, n->NyC)
                                                : mk_OX(qi.O, qi.X, qi.S, B, qi.dD, qi.P, qi.P, qi.Nx, qi.Ny, qi.Nx, qi.Ny)};
                RET_IF(do_contract_group(ctx, qs, 2, 2, 0));
                n->ox_done |= 1u << (l + 1);
                q.O_stale = true;
                continue;
            }
            hipError_t e;
            {
                Bracket br(ctx, KID_CONTRACT, ((double)k.R * k.K + (double)k.K * k.C + (double)k.R * k.C) * k.P * 8.0);
                e = launch_contract(bc(ctx, k), ctx->cur);
            }
            if (e == hipSuccess) { q.O_stale = true; continue; }
            if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "contract(compact decoder)", e);
            (void)hipGetLastError();
            // declined: from here down the full-grid decoder; the levels already done are expanded first
            n->compact = compact = false;
            for (int l2 = L - 2; l2 > l; --l2) {
                Pair& q2 = n->pr[l2];
                RET_IF(do_resize(ctx, q2.Oc, q2.O, (long)B * q2.dD, n->NxC, n->NyC, q2.Nx, q2.Ny));
                q2.O_stale = false;
            }
        }
        RET_IF(do_conv_up(ctx, in.O, q.F, q.p, q.O, B, q.dD, q.dM, q.Nx, q.Ny, in.Nx, in.Ny));
    }
    if (recon_d) {   // :1373 fft_inv of the up-sampled last output, fused zero-pad
        const bool nooverlap = flag(AEFFT_F_NOOVERLAP);
        const bool async = lazy && ctx->aux[0] != nullptr && !nooverlap && overlap_pays && !ctx->prof;
        n->recon_deferred = nullptr;
        if (async && n->input_ready && !flag(AEFFT_F_NODEFER)) {
            // pipelined loop (aefft_net_set_input_ready): launched by aefft_net_step_grad after the gradient half instead
            n->recon_deferred = recon_d;
            n->last_frames = frames_d; n->last_frames_u8 = ctx->in_u8;
            n->have_forward = true; n->have_grad = false;
            return AEFFT_OK;
        }
        if (async) {
            // training step: nothing downstream reads the reconstruction, so its (bandwidth-bound) inverse FFT runs on a side
            // stream underneath the latency-bound gradient contractions; aefft_net_step_grad joins it before returning
            if (!fork_recorded) HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[0], ctx->ev_fork, 0));
            ctx->cur = ctx->aux[0];
        }
        // (a side-stream transform has its own column/row workspace: the main stream's FFTs of non-pruned kernel supports use WS_MID)
        const int rc = launch_recon(n, recon_d, async ? WS_MID3 : WS_MID);
        ctx->cur = ctx->stream;
        RET_IF(rc);
        n->recon_pending = async;
    }
    n->last_frames = frames_d; n->last_frames_u8 = ctx->in_u8;
    n->have_forward = true; n->have_grad = false;
    return AEFFT_OK;
}

// Layer exports and bursts read per-frame spectra: after a training step in operator form the activation buffers hold operators,
// so the per-frame forward of the same frames is run first (with the CURRENT weights; the step's gradient state is kept).
static int ensure_frames(aefft_net* n)
{
    if (n->op_chain && n->act_stale) {
        // chain mode: X_l,b = A_l [x_b; 1], O_l,b = O^_l [x_b; 1] from the RESIDENT input spectra and the operators of the last
        // step_grad (set op_fwd: intact until the step after next's tail launch) -- neither the caller's frame buffer nor the
        // current (possibly updated) weights enter.  Hidden layers stay to be formed on request (H_stale).
        aefft_ctx* ctx = n->ctx;
        const Pair& q0 = n->pr[0];
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            hipError_t e = hipSuccess;
            if (l > 0) {
                Bracket br(ctx, KID_OPFORM, ((double)OPC * q.dD + (double)n->B * (q.dD + n->D)) * q.P * 8.0);
                e = launch_op_expand(q.opA[n->op_fwd], n->Xf, q.X, n->B, n->D, q.dD, q0.Nx, q0.Ny, q.Nx, q.Ny, ctx->cur);
            }
            if (e == hipSuccess) {
                Bracket br(ctx, KID_OPFORM, ((double)OPC * q.dD + (double)n->B * (q.dD + n->D)) * n->Pc * 8.0);
                e = launch_op_expand(q.opO[n->op_fwd], n->Xf, q.P != n->Pc ? q.Oc : q.O, n->B, n->D, q.dD, q0.Nx, q0.Ny, n->NxC, n->NyC, ctx->cur);
            }
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "op_expand", e);
            q.H_stale = true; q.O_stale = q.P != n->Pc;
        }
        n->pr[0].X = n->Xf;
        n->act_stale = false;
        return AEFFT_OK;
    }
    if (!n->op_state) return AEFFT_OK;
    // (operator form without the chain launch: the activation buffers hold the operators themselves; the per-frame forward of the
    // same frames is run -- the caller's frame buffer must still hold them, include/aefft.h)
    const bool hg = n->have_grad, u8 = n->ctx->in_u8;
    n->ctx->in_u8 = n->last_frames_u8;
    const int rcf = net_forward(n, n->last_frames, nullptr, false, false);
    n->ctx->in_u8 = u8;
    RET_IF(rcf);
    n->have_grad = hg;
    return AEFFT_OK;
}

extern "C" int aefft_net_forward(aefft_net* n, const float* frames_d, float* recon_d)
{
    RET_IF(net_forward(n, frames_d, recon_d, false));
    return mark_step_point(n);
}

extern "C" int aefft_net_get_layer(aefft_net* n, int layer, float* out_d, int* ch, int* nx, int* ny)
{
    if (!n || layer < 0 || layer > 4 * n->L) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_get_layer: bad layer index");
    aefft_ctx* ctx = n->ctx;
    RET_IF(join_recon(ctx));
    if (out_d && n->have_forward) RET_IF(ensure_frames(n));
    const int L = n->L, B = n->B;
    int c, x, y, xi, yi;            // channels, output size, stored spectrum size
    const float2* S = nullptr;
    if (layer == 0) { c = n->D; x = xi = n->Nx; y = yi = n->Ny; }
    else if (layer <= 2 * L) {
        const Pair& q = n->pr[(layer - 1) / 2];
        x = xi = q.Nx; y = yi = q.Ny;
        if (layer & 1) { c = q.dD; S = q.X; }
        else {
            c = q.dM; S = q.H;
            if (out_d && q.H_stale) {          // hidden layer skipped by the training step's forward: form it now (fft_backproplib.cu:1347)
                Pair& qm = n->pr[(layer - 1) / 2];
                if (n->upd_after_fwd && n->op_chain) {
                    // chain form after aefft_net_step_apply: X_l is the step's own (expanded from its operators), so the hidden layer must
                    // come from the step's encoder too, not from the updated one.  The update was w <- w - D with D left in the momentum
                    // buffer: c_old = c + Dc, b_old = b + Db (to one rounding of the subtraction), its spectrum into the pair's planar C
                    // buffer -- which this form keeps stale anyway (spectra_valid stays false: rebuilt from the current weights on demand).
                    const size_t nk = (size_t)qm.dM * qm.dD * qm.Nk * qm.Nl;
                    void *tmp, *real = nullptr;
                    RET_IF(ws_get(ctx, WS_TMP, sizeof(float) * (nk + qm.dM), &tmp));
                    float* c_old = (float*)tmp; float* b_old = c_old + nk;
                    hipError_t e = launch_vec_add(c_old, qm.c, qm.Dc, (long)nk, ctx->cur);
                    if (e == hipSuccess) e = launch_vec_add(b_old, qm.b, qm.Db, qm.dM, ctx->cur);
                    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "get_layer: previous encoder", e);
                    if (!pruned_supported(qm.Nk, qm.Nl, qm.Nx, qm.Ny)) real = n->real;
                    qm.spectra_valid = false;
                    RET_IF(do_pad_r2c(ctx, c_old, qm.C, (float*)real, (long)qm.dM * qm.dD, qm.Nx, qm.Ny, qm.Nk, qm.Nl));
                    RET_IF(do_conv(n->ctx, qm.X, qm.C, b_old, qm.H, n->B, qm.dM, qm.dD, qm.Nx, qm.Ny));
                } else {
                    RET_IF(ensure_spectra(n, qm));
                    RET_IF(do_conv(n->ctx, qm.X, qm.C, qm.b, qm.H, n->B, qm.dM, qm.dD, qm.Nx, qm.Ny));
                }
                qm.H_stale = false;
            }
        }
    } else {
        const int nn = (layer - 1) / 2;           // decoder conv index L..2L-1
        const Pair& q = n->pr[2 * L - 1 - nn];
        c = q.dD; S = q.O; xi = q.Nx; yi = q.Ny;
        if (q.O_stale) { S = q.Oc; xi = n->NxC; yi = n->NyC; }      // training-step forward: the layer is stored on its support only
        if (layer & 1) { x = q.Nx; y = q.Ny; } else { x = q.Nxin; y = q.Nyin; }   // odd: conv output; even: up-sampled
    }
    if (ch) *ch = c;
    if (nx) *nx = x;
    if (ny) *ny = y;
    if (!out_d) return AEFFT_OK;
    if (!n->have_forward) return fail(ctx, AEFFT_ESTATE, "aefft_net_get_layer: no forward pass yet");
    if (layer == 0) {
        if (n->last_frames_u8) {
            hipError_t e = launch_u8_to_f32(out_d, reinterpret_cast<const unsigned char*>(n->last_frames), (long)B * c * x * y, ctx->stream);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "get_layer: 8-bit frames", e);
            return AEFFT_OK;
        }
        HIPCHK(ctx, hipMemcpyAsync(out_d, n->last_frames, sizeof(float) * B * c * x * y, hipMemcpyDeviceToDevice, ctx->stream));
        return AEFFT_OK;
    }
    RET_IF(do_c2r(ctx, S, out_d, (long)B * c, xi, yi, x, y, 1.0f / ((float)x * (float)y)));
    return mark_step_point(n);
}

extern "C" int aefft_magnitude(aefft_ctx* ctx, const float* X_d, float* mag_d, long planes, int ch, int Nx, int Ny, int shift)
{
    if (!ctx || !X_d || !mag_d || planes <= 0 || ch <= 0 || Nx <= 0 || Ny <= 1) return fail(ctx, AEFFT_EINVAL, "aefft_magnitude: bad argument");
    Bracket br(ctx, KID_RESIZE, (double)planes * ((double)Nx * (Ny / 2 + 1) * 8.0 + (double)Nx * Ny * 4.0));
    hipError_t e = launch_magnitude(CF2(X_d), mag_d, planes, ch, Nx, Ny, shift, ctx->stream);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "magnitude", e);
    return AEFFT_OK;
}

extern "C" int aefft_net_layers_layout(aefft_net* n, size_t* offsets_h)
{
    if (!n || !offsets_h) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_layers_layout: bad argument");
    size_t off = 0;
    for (int l = 0; l <= 4 * n->L; ++l) {
        int c, x, y;
        RET_IF(aefft_net_get_layer(n, l, nullptr, &c, &x, &y));
        offsets_h[l] = off;
        off += (size_t)n->B * c * x * y;
    }
    offsets_h[4 * n->L + 1] = off;
    return AEFFT_OK;
}

// All layers of the last forward in coordinate space (fft_l = 1, fft_backproplib.cu:1347,1357,1361).  A decoder conv output and
// the up-sampled layer after it are inverse transforms of the SAME spectrum onto two grids; an encoder's pooled input and the
// previous hidden layer likewise -- each stored spectrum is read where it lies, the crop / zero-pad is fused into the transform.
extern "C" int aefft_net_get_layers(aefft_net* n, float* out_d)
{
    if (!n || !out_d) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_get_layers: bad argument");
    if (!n->have_forward) return fail(n->ctx, AEFFT_ESTATE, "aefft_net_get_layers: no forward pass yet");
    std::vector<size_t> off(4 * n->L + 2);
    RET_IF(aefft_net_layers_layout(n, off.data()));
    for (int l = 0; l <= 4 * n->L; ++l) RET_IF(aefft_net_get_layer(n, l, out_d + off[l], nullptr, nullptr, nullptr));
    return AEFFT_OK;
}

// expand a decoder output that the training-step forward kept on its support only
static int ensure_O(aefft_net* n, Pair& q)
{
    if (!q.O_stale) return AEFFT_OK;
    RET_IF(do_resize(n->ctx, q.Oc, q.O, (long)n->B * q.dD, n->NxC, n->NyC, q.Nx, q.Ny));
    q.O_stale = false;
    return AEFFT_OK;
}

// gradient half of one loop-body iteration on pair q: needs X (= T, autoencoder.cpp:194) and the current O.
static int pair_grad(aefft_net* n, Pair& q)
{
    aefft_ctx* ctx = n->ctx;
    float* g = n->grad + q.goff;
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    RET_IF(ensure_O(n, q));
    RET_IF(do_gradient(ctx, q.X, q.X, q.O, q.C, q.F, q.b, q.S, q.dc, q.df, g + 2 * nk, g + 2 * nk + q.dM, n->B, q.dM, q.dD, q.Nx, q.Ny));
    const long planes = (long)q.dM * q.dD;
    if (q.part) return do_c2r_shrink(ctx, q.dc, g, nullptr, q.part, 2 * planes, q.Nx, q.Ny, q.Nk, q.Nl);   // dc|df -> dck|dfk, one launch
    RET_IF(do_c2r_shrink(ctx, q.dc, g, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl));
    return do_c2r_shrink(ctx, q.df, g + nk, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl);
}

// update half: weights, new spectra, re-forward of the pair alone, post-update MSE accumulated into *mse_slot (pre-zeroed)
static int pair_apply(aefft_net* n, Pair& q, float del, int maxdiff, int sym, float gscale, float* mse_slot)
{
    aefft_ctx* ctx = n->ctx;
    float* g = n->grad + q.goff;
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    RET_IF(do_update(ctx, q.c, q.f, q.b, q.p, g, g + nk, g + 2 * nk, g + 2 * nk + q.dM, Momentum{q.Dc, q.Df, q.Db, q.Dp},
                     q.dM, q.dD, q.Nk, q.Nl, del, maxdiff, sym, gscale, n->burst ? nullptr : mse_slot));
    RET_IF(pair_spectra(n, q));
    // re-forward of this pair alone (fft_backproplib.cu:1460-1461) and its MSE (:1463)
    RET_IF(do_conv(ctx, q.X, q.C, q.b, q.H, n->B, q.dM, q.dD, q.Nx, q.Ny));
    RET_IF(do_conv(ctx, q.H, q.F, q.p, q.O, n->B, q.dD, q.dM, q.Nx, q.Ny));
    if (mse_slot) RET_IF(do_diff_mse(ctx, q.X, q.O, nullptr, mse_slot, nullptr, n->B, q.dM, q.dD, q.Nx, q.Ny));
    return AEFFT_OK;
}

extern "C" int aefft_net_train_pair(aefft_net* n, int l, int n_iter, float del0, int maxdiff, int sym, float* mse_h)
{
    if (!n || l < 0 || l >= n->L || n_iter < 0) return fail(n ? n->ctx : nullptr, AEFFT_EINVAL, "aefft_net_train_pair: bad argument");
    n->upd_after_fwd = false;
    aefft_ctx* ctx = n->ctx;
    if (!n->have_forward) return fail(ctx, AEFFT_ESTATE, "aefft_net_train_pair: run aefft_net_forward first (the burst trains on its layers)");
    RET_IF(mse_flush(n));
    RET_IF(join_recon(ctx));
    RET_IF(ensure_frames(n));
    Pair& q = n->pr[l];
    RET_IF(ensure_spectra(n, q));
    RET_IF(ensure_O(n, q));
    if ((size_t)(n_iter + 1) > n->mse_cap) {
        float* nm;
        RET_IF(net_alloc_t(n, &nm, (size_t)n_iter + 1));
        n->mse_dev = nm; n->mse_cap = (size_t)n_iter + 1;
    }
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    // momentum lives only inside the burst (fft_backproplib.cu:1420-1423)
    HIPCHK(ctx, hipMemsetAsync(q.Dc, 0, nk * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(q.Df, 0, nk * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(q.Db, 0, q.dM * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(q.Dp, 0, q.dD * 4, ctx->stream));
    const float del = 0.1f * del0;                      // :1445
    HIPCHK(ctx, hipMemsetAsync(n->mse_dev, 0, sizeof(float) * (n_iter + 1), ctx->stream));
    RET_IF(do_diff_mse(ctx, q.X, q.O, nullptr, n->mse_dev, nullptr, n->B, q.dM, q.dD, q.Nx, q.Ny));     // :1440
    n->burst = true;
    int rcb = AEFFT_OK;
    for (int it = 0; it < n_iter && rcb == AEFFT_OK; ++it) {
        rcb = pair_grad(n, q);
        if (rcb == AEFFT_OK) rcb = pair_apply(n, q, del, maxdiff, sym, 1.0f, n->mse_dev + it + 1);
    }
    n->burst = false;
    RET_IF(rcb);
    n->have_grad = false;
    if (mse_h) {
        HIPCHK(ctx, hipMemcpyAsync(mse_h, n->mse_dev, sizeof(float) * (n_iter + 1), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    q.G_valid = false; n->packed_valid = false; n->chain_valid = false;         // the burst changed this pair's weights (and used S)
    return mark_step_point(n);
}

// Step mode runs the same per-pair sequences as pair_grad / pair_apply, phase by phase over ALL pairs, so that
// the independent contractions of a phase (4 x S, 4 x dc + 4 x df, 4 + 4 re-forward convs) go out as one launch each.
static int bias_and_kgrad(aefft_net* n, Pair& q)
{
    RET_IF(ensure_O(n, q));
    aefft_ctx* ctx = n->ctx;
    float* g = n->grad + q.goff;
    const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
    const float norm = (float)q.Nx * (float)q.Ny, Norm = grad_norm(q.dM, q.dD, q.Nx, q.Ny);
    {
        Bracket br(ctx, KID_BIASGRAD, ((double)(q.dM * q.dD + q.dM + q.dD) + 2.0 * n->B * q.dD) * 8.0);
        hipError_t e = launch_bias_grad(q.O, q.X, q.F, q.b, q.df, g + 2 * nk, g + 2 * nk + q.dM, n->B, q.dM, q.dD, q.P, norm, Norm, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "bias_grad", e);
    }
    const long planes = (long)q.dM * q.dD;
    if (q.part) return do_c2r_shrink(ctx, q.dc, g, nullptr, q.part, 2 * planes, q.Nx, q.Ny, q.Nk, q.Nl);
    RET_IF(do_c2r_shrink(ctx, q.dc, g, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl));
    return do_c2r_shrink(ctx, q.df, g + nk, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl);
}

// The slot sums of the last step's post-update MSE when aefft_net_step_apply was told not to deliver them (mse_d == NULL): they ride as a
// trailing workgroup of the next step's gradient launch (grads_grouped); anything else that needs them first calls this.
static int mse_flush(aefft_net* n)
{
    if (!n->mse_pending) return AEFFT_OK;
    aefft_ctx* ctx = n->ctx;
    Bracket br(ctx, KID_DIFFMSE, 4.0 * n->L * MSE_SLOTS);
    hipError_t e = launch_mse_finish(n->mse_slots, n->mse_post, nullptr, n->L, ctx->cur, nullptr, n->grad + n->grad_n, n->mse_pending_scale);
    if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "mse_finish(deferred)", e);
    n->mse_pending = false;
    return AEFFT_OK;
}

static int grads_grouped(aefft_net* n)
{
    aefft_ctx* ctx = n->ctx;
    Contract qs[8];
    const bool op = op_mode(n);
    if (op) {
        // the batch moments, S_l = sum_b (O_b - X_b) X_b^H and the DC error sums of every pair from the operators: one launch
        SgradGroup sg{};
        double bytes = ((double)n->B * n->D + (double)OPC * OPC) * n->pr[0].P * 8.0;      // the input spectra in, the moments out
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            const OpView ov = op_view(n, l);
            const int nxo = ov.nxo, nyo = ov.nyo;
            sg.q[l] = OpPair{ov.A, ov.O, q.S, q.es, q.dD, q.Nx, q.Ny, nxo, nyo, q.P, bins(nxo, nyo)};
            bytes += ((double)OPC * q.dD * (q.P + bins(nxo, nyo)) + (double)q.dD * q.dD * q.P) * 8.0;
        }
        sg.n = n->L; sg.Xf = n->Xf; sg.Mout = n->Mhat; sg.B = n->B; sg.D0 = n->D; sg.Nx0 = n->pr[0].Nx; sg.Ny0 = n->pr[0].Ny; sg.P0 = n->pr[0].P;
        Bracket br(ctx, KID_SGRAD, bytes);
        hipError_t e = launch_msgrad_group(sg, ctx->cur);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "sgrad", e);
    }
    bool comp = false;
    for (int l = 0; l < n->L; ++l) comp = comp || n->pr[l].O_stale;
    for (int l0 = 0; l0 < n->L && !op; l0 += 4) {
        const int m = std::min(4, n->L - l0);
        if (!comp) {
            for (int i = 0; i < m; ++i) { Pair& q = n->pr[l0 + i]; qs[i] = mk_S(q.X, q.X, q.O, q.S, n->B, q.dD, q.P); }
            RET_IF(do_contract_group(ctx, qs, m, m, 1));
            continue;
        }
        if (!n->xx_done) {
            for (int i = 0; i < m; ++i) { Pair& q = n->pr[l0 + i]; qs[i] = mk_XXneg(q.X, q.S, n->B, q.dD, q.P); }
            RET_IF(do_contract_group(ctx, qs, m, m, 1));
        }
        int mo = 0;
        for (int i = 0; i < m; ++i) {
            Pair& q = n->pr[l0 + i];
            if (n->xx_done && (n->ox_done >> (l0 + i) & 1u)) continue;          // rode along with a decoder launch of the forward
            qs[mo++] = q.O_stale ? mk_OX(q.Oc, q.X, q.S, n->B, q.dD, q.P, n->Pc, q.Nx, q.Ny, n->NxC, n->NyC)
                                 : mk_OX(q.O, q.X, q.S, n->B, q.dD, q.P, q.P, q.Nx, q.Ny, q.Nx, q.Ny);
        }
        if (mo == 1) RET_IF(do_contract(ctx, qs[0]));
        else if (mo > 1) RET_IF(do_contract_group(ctx, qs, mo, mo, 1));
    }
    n->xx_done = false; n->ox_done = 0;
    // DC-bin terms and the pruned inverse transforms of all pairs: one launch each when the pairs share (Nk, Nl)
    const bool nogroup = flag(AEFFT_F_NOGROUP);
    bool same = op || (n->L > 1 && n->L <= 8 && !nogroup);
    for (int l = 0; l < n->L && same; ++l) {
        const Pair& q = n->pr[l];
        same = q.Nk == n->pr[0].Nk && q.Nl == n->pr[0].Nl && pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny);
    }
    const bool noq = flag(AEFFT_F_NOQPATH);
    bool qpath = op || (same && !noq && n->pr[0].Nk == n->pr[0].Nl && (n->pr[0].Nk == 3 || n->pr[0].Nk == 5));
    for (int l = 0; l < n->L && qpath; ++l) qpath = n->pr[l].Q != nullptr;
    if (same) {
        BiasGradGroup bg{};
        PrunedGroup pg{};
        WgradGroup wg{};
        double bbytes = 0, kbytes = 0, wbytes = 0;
        const int T = 2 * n->pr[0].Nk - 1;
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            float* g = n->grad + q.goff;
            const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
            const float Norm = grad_norm(q.dM, q.dD, q.Nx, q.Ny);
            bg.a[l] = BiasGradArgs{q.O_stale ? q.Oc : q.O, q.X, q.F, q.b, qpath ? nullptr : q.df, g + 2 * nk, g + 2 * nk + q.dM, n->B, q.dM, q.dD, q.P,
                                   (float)q.Nx * (float)q.Ny, Norm, q.O_stale ? n->Pc : q.P, qpath ? q.es : nullptr, op ? q.es : nullptr};
            if (!q.spectra_valid) {
                // (operator form: the step left the planar spectra stale; F at the DC bin is record 0 of the
                // bin-major copy -- element (d1*dM + m) of the pair's F segment, stride 1.  Only F is read through P in this form.)
                if (!(op && qpath && n->Wp && n->packed_valid)) return fail(ctx, AEFFT_ESTATE, "gradient: stale kernel spectra");
                bg.a[l].F = n->Wp + n->pack.seg[2 * n->L - 1 - l].off;      // (segments: C_0 .. C_{L-1}, F_{L-1} .. F_0)
                bg.a[l].P = 1;
            }
            bbytes += ((double)(q.dM * q.dD + q.dM + q.dD) + 2.0 * n->B * q.dD) * 8.0;
            if (qpath) {
                // weight gradients through Q = pruned inverse transform of S on the (2Nk-1)^2 offsets (weight_kernels.hip): no dc|df spectra
                pg.q[l] = PrunedProb{q.S, q.Q, (long)q.dD * q.dD, q.Nx, q.Ny, 1.0f};
                pg.chunks[l] = q.Qn;
                wg.q[l] = WgradProb{q.c, q.f, q.Q, q.es, q.b, g, g + nk, q.dM, q.dD, 1.0f / (Norm * (float)n->B), (float)q.Nx * (float)q.Ny, 1};
                kbytes += (double)q.dD * q.dD * (q.P * 8.0 + T * T * 4.0);
                wbytes += (2.0 * nk + (double)q.dD * q.dD * T * T) * 4.0 + 2.0 * nk * 4.0;
            } else {
                pg.q[l] = PrunedProb{q.dc, g, 2L * q.dM * q.dD, q.Nx, q.Ny, 1.0f};
                kbytes += 2.0 * q.dM * q.dD * (q.P * 8.0 + q.Nk * q.Nl * 4.0);
            }
        }
        bg.n = pg.n = wg.n = n->L;
        if (!qpath) {
            // dc | df of every pair in one launch (8 problems)
            for (int l0 = 0; l0 < n->L; l0 += 4) {
                const int m = std::min(4, n->L - l0);
                for (int i = 0; i < m; ++i) {
                    Pair& q = n->pr[l0 + i];
                    const float Norm = grad_norm(q.dM, q.dD, q.Nx, q.Ny);
                    qs[i] = mk_dc(q.F, q.S, q.dc, n->B, q.dM, q.dD, q.P, Norm);
                    qs[m + i] = mk_df(q.C, q.S, q.df, n->B, q.dM, q.dD, q.P, Norm);
                }
                RET_IF(do_contract_group(ctx, qs, 2 * m, m, 2));
            }
        }
        if (!qpath) {
            Bracket br(ctx, KID_BIASGRAD, bbytes);
            hipError_t e = launch_bias_grad_group(bg, ctx->cur);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "bias_grad(group)", e);
        }
        hipError_t e;
        {
            Bracket br(ctx, KID_KGRAD, kbytes + (qpath ? bbytes : 0.0));
            e = qpath ? launch_kgrad_group_taps(pg, ctx->tw, T, ctx->cur, &bg)      // (the DC-bin terms ride along as extra workgroups)
                      : launch_kgrad_group(pg, ctx->tw, n->pr[0].Nk, n->pr[0].Nl, ctx->cur);
        }
        if (e == hipSuccess && qpath) {
            for (int l = 0; l < n->L; ++l) wg.q[l].nq = pg.chunks[l];
            if (n->mse_pending) {      // the previous step's MSE sums: one more workgroup of this launch (they reach the packed buffer's tail before the all-reduce)
                wg.fin_slots = n->mse_slots; wg.fin_out = n->mse_post; wg.fin_tail = n->grad + n->grad_n; wg.fin_L = n->L; wg.fin_scale = n->mse_pending_scale;
            }
            Bracket br(ctx, KID_WGRAD, wbytes);
            e = launch_wgrad_taps_group(wg, n->pr[0].Nk, ctx->cur);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "wgrad(group)", e);
            n->mse_pending = false;
            return AEFFT_OK;
        }
        if (e == hipSuccess) return AEFFT_OK;
        if (e != hipErrorInvalidValue || qpath) return fail(ctx, AEFFT_EHIP, "kgrad(group)", e);
        (void)hipGetLastError();
        for (int l = 0; l < n->L; ++l) {          // bias terms are done; only the transforms pair by pair
            Pair& q = n->pr[l];
            float* g = n->grad + q.goff;
            const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
            const long planes = (long)q.dM * q.dD;
            if (q.part) { RET_IF(do_c2r_shrink(ctx, q.dc, g, nullptr, q.part, 2 * planes, q.Nx, q.Ny, q.Nk, q.Nl)); continue; }
            RET_IF(do_c2r_shrink(ctx, q.dc, g, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl));
            RET_IF(do_c2r_shrink(ctx, q.df, g + nk, n->real, nullptr, planes, q.Nx, q.Ny, q.Nk, q.Nl));
        }
        return AEFFT_OK;
    }
    // pairs with different kernel supports: dc | df per group of pairs, then pair by pair
    for (int l0 = 0; l0 < n->L; l0 += 4) {
        const int m = std::min(4, n->L - l0);
        for (int i = 0; i < m; ++i) {
            Pair& q = n->pr[l0 + i];
            const float Norm = grad_norm(q.dM, q.dD, q.Nx, q.Ny);
            qs[i] = mk_dc(q.F, q.S, q.dc, n->B, q.dM, q.dD, q.P, Norm);
            qs[m + i] = mk_df(q.C, q.S, q.df, n->B, q.dM, q.dD, q.P, Norm);
        }
        RET_IF(do_contract_group(ctx, qs, 2 * m, m, 2));
    }
    for (int l = 0; l < n->L; ++l) RET_IF(bias_and_kgrad(n, n->pr[l]));
    return AEFFT_OK;
}

// post-update MSE of pair q on the current frames (fft_backproplib.cu:1460-1463).  Step mode never reads the
// re-forward's H and O again (the next forward overwrites them), so they are not materialised: G = F.C per bin
// (into the dead S workspace), then one pass over X with the MSE epilogue.  Falls back to conv, conv, diff_mse
// for shapes the lean kernel does not serve (dD == 1 or B == 1).
static int reforward_mse(aefft_net* n, Pair& q, float* mse_slots, bool* g_left_in_S = nullptr)
{
    if (g_left_in_S) *g_left_in_S = false;
    aefft_ctx* ctx = n->ctx;
    const bool nofuse = flag(AEFFT_F_NOFUSEMSE);
    if (!nofuse && q.dD >= 2 && n->B >= 2) {
        RET_IF(do_contract(ctx, mk_G(q.F, q.C, q.G, q.dM, q.dD, q.P)));
        const Contract m = mk_gmse(q.G, q.X, q.F, q.b, q.p, mse_slots, n->B, q.dM, q.dD, q.Nx, q.Ny);
        hipError_t e;
        {
            Bracket br(ctx, KID_CONTRACT, contract_bytes(m));
            e = launch_contract(m, ctx->cur);
        }
        if (e == hipSuccess) { if (g_left_in_S) *g_left_in_S = true; return AEFFT_OK; }
        if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "contract(mse)", e);
        (void)hipGetLastError();
    }
    RET_IF(join_recon(ctx));                                                  // a deferred reconstruction may still be reading q.O (== Oc when P == Pc)
    RET_IF(do_conv(ctx, q.X, q.C, q.b, q.H, n->B, q.dM, q.dD, q.Nx, q.Ny));   // :1460
    RET_IF(do_conv(ctx, q.H, q.F, q.p, q.O, n->B, q.dD, q.dM, q.Nx, q.Ny));   // :1461
    return do_diff_mse(ctx, q.X, q.O, nullptr, n->mse_post + (&q - n->pr.data()), nullptr, n->B, q.dM, q.dD, q.Nx, q.Ny);   // :1463
}

static int apply_grouped(aefft_net* n, float del, int maxdiff, int sym, float gscale, float* mse_d)
{
    aefft_ctx* ctx = n->ctx;
    for (auto& q : n->pr) q.G_valid = false;          // the weights are about to change
    n->packed_valid = false; n->chain_valid = false;
    const bool nogroup1 = flag(AEFFT_F_NOGROUP);
    bool fused_upd = false;                                                // the tap half of the update rides in the tail launch (below)
    bool gp_route = false;                                                 // the spectra launch wrote G' = F'.C'/(dM dD) for every pair but the innermost
    UpdateGroup wupd{};
    bool grouped_w = n->L > 1 && n->L <= 8 && !nogroup1;
    for (int l = 0; l < n->L && grouped_w; ++l) {
        const Pair& q = n->pr[l];
        grouped_w = q.Nk == n->pr[0].Nk && q.Nl == n->pr[0].Nl && pruned_supported(q.Nk, q.Nl, q.Nx, q.Ny);
    }
    // multiobjective terms (fft_backproplib.cu:709-753) of every pair in one grouped launch; their outputs and the chunk partial sums
    // live in per-net buffers (allocated the first time maxdiff is asked for)
    GdiffGroup gd{};
    if (grouped_w && maxdiff) {
        const int kl = n->pr[0].Nk * n->pr[0].Nl;
        grouped_w = kl == 9 || kl == 25 || kl == 49;
        if (grouped_w && !n->gd_out) {
            size_t no = 0, np_ = 0;
            for (const Pair& q : n->pr) { no += 2 * (size_t)q.dM * q.dD * kl + q.dM + q.dD; np_ += gradient_diff_ws_floats(q.dM, q.dD, q.Nk, q.Nl); }
            RET_IF(net_alloc_t(n, &n->gd_out, no));
            RET_IF(net_alloc_t(n, &n->gd_part, np_));
        }
        if (grouped_w) {
            float *o = n->gd_out, *pw = n->gd_part;
            double gbytes = 0;
            for (int l = 0; l < n->L; ++l) {
                Pair& q = n->pr[l];
                const size_t nk = (size_t)q.dM * q.dD * kl;
                gd.q[l] = GdiffProb{q.c, q.f, q.b, q.p, o, o + nk, o + 2 * nk, o + 2 * nk + q.dM, pw, q.dM, q.dD, 0, 0};
                o += 2 * nk + q.dM + q.dD; pw += gradient_diff_ws_floats(q.dM, q.dD, q.Nk, q.Nl);
                gbytes += (double)nk * 16.0;
            }
            gd.n = n->L;
            Bracket br(ctx, KID_GDIFF, gbytes);
            hipError_t e = launch_gradient_diff_group(gd, n->pr[0].Nk, n->pr[0].Nl, ctx->cur);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "gradient_diff(group)", e);
        }
    }
    if (grouped_w) {
        UpdateGroup ug{};
        PrunedGroup pg{};
        double ubytes = 0, kbytes = 0;
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            float* g = n->grad + q.goff;
            const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
            ug.a[l] = mk_update(q.c, q.f, q.b, q.p, g, g + nk, g + 2 * nk, g + 2 * nk + q.dM, Momentum{q.Dc, q.Df, q.Db, q.Dp},
                                q.dM, q.dD, q.Nk, q.Nl, del, sym, gscale, n->mse_post + l);
            if (maxdiff) { ug.a[l].cd = gd.q[l].cd; ug.a[l].fd = gd.q[l].fd; ug.a[l].bd = gd.q[l].bd; ug.a[l].pd = gd.q[l].pd; }
            ubytes += (double)nk * 4.0 * 8;
        }
        ug.n = n->L;
        const bool ride = op_mode(n) && n->Wp != nullptr;                  // the bin-major copy for the next step's chain: same taps, same launch
        // Fused update (operator form, plain gradients): no update launch.  The spectra launch reads every tap THROUGH the pending
        // update (w - clip_step(g, D): TapUpd) and carries the bias half as a trailing workgroup per pair; the taps and their momentum
        // are stored in place by trailing workgroups of the tail launch (tail_kernel) -- nothing in between reads them.
        fused_upd = ride && !sym && !maxdiff && !flag(AEFFT_F_NOFUSEUPD) && !ctx->prof;
        // Operator form with the chain launch: NO planar spectra are written.  The next step's chain reads the bin-major record Wp
        // and the compact planes Cc_l (C_l where the next pair's grid lands); the post-update MSE reads G'_l = F'_l.C'_l/(dM dD) --
        // dD*dD planes per pair, the spectrum of the (2Nk-1)^2 kernel f' (*) c' whose taps the transforming workgroups form
        // themselves (gspec_gbody) -- and the innermost pair from Wp.  Planar C|F are formed when something else asks (ensure_spectra).
        gp_route = ride && n->pr[0].Cc != nullptr && 2 * (n->L - 1) <= 8 && chain_switches_ok() && n->compact &&
                   n->pr[n->L - 1].P == n->pack.Pc && n->pr[n->L - 1].dD <= CH_VMAX && n->pr[n->L - 1].dM <= CH_VMAX;
        BiasUpdGroup bu{};
        TapUpd tu[8] = {};
        if (fused_upd) {
            for (int l = 0; l < n->L; ++l) {
                Pair& q = n->pr[l];
                float* g = n->grad + q.goff;
                const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
                tu[l] = TapUpd{g, q.Dc, ug.a[l].del, ug.a[l].alpha, ug.a[l].gscale};        // (c|f, dck|dfk, Dc|Df: each pair contiguous)
                bu.a[l] = BiasUpd{q.b, q.p, q.Db, q.Dp, g + 2 * nk, g + 2 * nk + q.dM, n->mse_post + l, q.dM, q.dD};
            }
            bu.n = n->L; bu.del = ug.a[0].del; bu.alpha = ug.a[0].alpha; bu.gscale = ug.a[0].gscale;
            n->pack.upd = 1; n->pack.upd_del = ug.a[0].del; n->pack.upd_alpha = ug.a[0].alpha; n->pack.upd_gscale = ug.a[0].gscale;
            wupd = ug;
        } else {
            n->pack.upd = 0;
            Bracket br(ctx, KID_UPDATE, ubytes);
            hipError_t e = launch_update_group(ug, ctx->cur);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "update(group)", e);
        }
        bool skip_inner = false;
        if (gp_route) {
            int k = 0;
            for (int l = 0; l + 1 < n->L; ++l) {
                Pair& q = n->pr[l];
                pg.q[k] = PrunedProb{nullptr, q.G, (long)q.dD * q.dD, q.Nx, q.Ny, 1.0f, 0, 0};
                pg.gsrc[k] = GtapSrc{q.c, q.f, q.dM, q.dD, 1.0f / ((float)q.dM * (float)q.dD)};
                pg.upd[k] = tu[l];
                kbytes += (double)q.dD * q.dD * q.P * 8.0 + 2.0 * q.dM * q.dD * q.Nk * q.Nl * 4.0;
                ++k;
            }
            const int k0 = k;
            k = cc_problems(n, pg, k0, &kbytes);
            for (int l = 0; l + 1 < n->L; ++l) pg.upd[k0 + l] = tu[l];
            pg.n = k;
        } else {
            for (int l = 0; l < n->L; ++l) {
                Pair& q = n->pr[l];
                pg.q[l] = PrunedProb{q.c, q.C, 2L * q.dM * q.dD, q.Nx, q.Ny, 1.0f};
                pg.upd[l] = tu[l];
                kbytes += 2.0 * q.dM * q.dD * (q.P * 8.0 + q.Nk * q.Nl * 4.0);
            }
            pg.n = n->L;
            // Without the compact planes: the innermost pair's PLANAR spectra are not written (the chain, the post-update MSE and the
            // DC-bin gradient terms take that pair from the bin-major record)
            skip_inner = ride && n->L > 1 && n->pr[n->L - 1].P == n->pack.Pc && n->pr[n->L - 1].dD <= CH_VMAX && n->pr[n->L - 1].dM <= CH_VMAX &&
                         !flag(AEFFT_F_NOCHAIN) && !flag(AEFFT_F_NOFUSEUPD) && !ctx->prof;
            if (skip_inner) {
                const Pair& qi = n->pr[n->L - 1];
                pg.n = n->L - 1;
                kbytes -= 2.0 * qi.dM * qi.dD * (qi.P * 8.0 + qi.Nk * qi.Nl * 4.0);
            }
        }
        hipError_t e;
        {
            Bracket br(ctx, KID_KSPEC, kbytes + ((op_mode(n) && n->Wp) ? (double)n->pack.Pc * n->pack.E * 8.0 : 0.0));
            e = launch_kspec_group(pg, ctx->tw, n->pr[0].Nk, n->pr[0].Nl, ctx->cur, ride ? &n->pack : nullptr, fused_upd ? &bu : nullptr);
            if (e == hipSuccess && ride) n->packed_valid = true;
            if (e == hipSuccess && skip_inner) n->pr[n->L - 1].spectra_valid = false;
            if (e == hipSuccess && gp_route) for (auto& q : n->pr) q.spectra_valid = false;
            if (e == hipSuccess && !gp_route) for (int l = 0; l < pg.n; ++l) n->pr[l].spectra_valid = true;
        }
        n->pack.upd = 0;
        if (e != hipSuccess) {
            if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "kspec(group)", e);
            (void)hipGetLastError();
            gp_route = false;
            if (fused_upd) {                                               // declined before anything ran: the separate update after all
                fused_upd = false;
                hipError_t e2 = launch_update_group(ug, ctx->cur);
                if (e2 != hipSuccess) return fail(ctx, AEFFT_EHIP, "update(group)", e2);
            }
            for (int l = 0; l < n->L; ++l) RET_IF(pair_spectra(n, n->pr[l]));
            for (auto& q : n->pr) q.spectra_valid = true;
        }
    } else for (int l = 0; l < n->L; ++l) {
        Pair& q = n->pr[l];
        float* g = n->grad + q.goff;
        const size_t nk = (size_t)q.dM * q.dD * q.Nk * q.Nl;
        RET_IF(do_update(ctx, q.c, q.f, q.b, q.p, g, g + nk, g + 2 * nk, g + 2 * nk + q.dM, Momentum{q.Dc, q.Df, q.Db, q.Dp},
                         q.dM, q.dD, q.Nk, q.Nl, del, maxdiff, sym, gscale, n->mse_post + l));
        RET_IF(pair_spectra(n, q));
        q.spectra_valid = true;
    }
    if (op_mode(n) && n->Wp) RET_IF(ensure_packed(n));     // the next step's chain reads the bin-major copy of the NEW weights
    bool g_taps = false;                                   // G' of EVERY pair at hand (operator form without the chain launch, HBM-sized spectra)
    if (op_mode(n) && !gp_route) {
        // HBM-sized kernel spectra (no pooling): the post-update MSE would read all 2*dM*dD planes of C'|F' back (6 GB at cfg3-P1).  G' =
        // F'.C'/(dM dD) as the spectrum of the (2Nk-1)^2-tap kernel f' (*) c' (weight_kernels.hip) is dD*dD planes written and read once.
        double cf_bytes = 0;
        for (int l = 0; l < n->L; ++l) cf_bytes += 2.0 * n->pr[l].dM * n->pr[l].dD * n->pr[l].P * 8.0;
        const bool want = (cf_bytes > 256e6 || flag(AEFFT_F_GTAPS)) && !fused_upd /* the taps are stored */ && !flag(AEFFT_F_NOQPATH);
        if (want) RET_IF(gprime_from_taps(n, &g_taps));
    }
    if (op_mode(n)) {
        // post-update MSE (fft_backproplib.cu:1460-1463) in operator form: R = A - F'(C' A / dM + b^) / dD - p^ per bin, then
        // sum_a R[a] M^ R[a]^H; the updated spectra (or their product G') are read once, nothing is stored (opform_kernels.hip)
        OpMseGroup og{};
        double bytes = 0;
        const bool inner_packed = n->Wp && n->packed_valid && n->pr[n->L - 1].P == n->pack.Pc;
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            const float scale = 1.0f / ((float)(2 * q.dM) * (float)q.Nx * (float)q.Ny * (float)n->B) / ((float)q.dD * q.Nx * q.Ny);   // as mk_gmse
            OpMsePair o{};
            o.A = op_view(n, l).A; o.C = q.C; o.F = q.F; o.b = q.b; o.p = q.p;
            o.slots = n->mse_slots + (size_t)l * MSE_SLOTS * MSE_SLOT_STRIDE;
            o.dD = q.dD; o.dM = q.dM; o.Nx = q.Nx; o.Ny = q.Ny; o.P = q.P; o.scale = scale;
            if (gp_route && l + 1 < n->L) {
                o.G = q.G; o.Fdc = n->Wp + n->pack.seg[2 * n->L - 1 - l].off; o.fdc_stride = 1;      // (F' at the DC bin: record 0 of the bin-major copy)
                bytes += ((double)q.dD * q.dD + (double)OPC * q.dD + (double)OPC * OPC) * q.P * 8.0;
            } else if (g_taps && !(l == n->L - 1 && inner_packed)) {
                RET_IF(ensure_spectra(n, q));
                o.G = q.G; o.Fdc = q.F; o.fdc_stride = q.P;                                            // (the planar F', DC bin)
                bytes += ((double)q.dD * q.dD + (double)OPC * q.dD + (double)OPC * OPC) * q.P * 8.0;
            } else {
                if (!(l == n->L - 1 && inner_packed)) RET_IF(ensure_spectra(n, q));
                bytes += (2.0 * q.dM * q.dD + (double)OPC * q.dD + (double)OPC * OPC) * q.P * 8.0;
            }
            og.q[l] = o;
        }
        og.n = n->L; og.Mhat = n->Mhat; og.Nx0 = n->pr[0].Nx; og.Ny0 = n->pr[0].Ny; og.P0 = n->pr[0].P;
        if (inner_packed) {      // the innermost pair reads the bin-major copy the kspec launch just refreshed
            og.Wp = n->Wp; og.E = n->pack.E;
            og.offC = n->pack.seg[n->L - 1].off; og.offF = n->pack.seg[n->L].off;
        }
        // The NEXT step's operator chain depends on the updated weights only (the record Wp and the planes Cc the spectra launch has just
        // written): it shares this launch, writing the other set of operator buffers, and the next aefft_net_step_grad starts from it.
        const bool ahead = n->op_chain && n->Wp && n->packed_valid && chain_switches_ok() && (n->compact || n->L == 1) && !flag(AEFFT_F_NOAHEAD);
        ChainArgs ca{};
        if (ahead) fill_chain(n, ca, n->op_set ^ 1, &bytes);
        {
            Bracket br(ctx, KID_OPMSE, bytes);
            hipError_t e = launch_opmse_group(og, ctx->cur, ahead ? &ca : nullptr, fused_upd ? &wupd : nullptr);
            if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "opmse", e);
        }
        if (ahead) { n->op_set ^= 1; n->chain_valid = true; }
        if (!mse_d && !ctx->prof && !flag(AEFFT_F_NOLAZYMSE)) {
            // nobody asked for the sums now: they are formed by one more workgroup of the next step's gradient launch (before its
            // all-reduce), by aefft_net_last_mse, or by whatever needs the slots next -- not by a launch of their own
            n->mse_pending = true; n->mse_pending_scale = gscale;
            return AEFFT_OK;
        }
        Bracket br(ctx, KID_DIFFMSE, 4.0 * n->L * MSE_SLOTS);
        hipError_t e = launch_mse_finish(n->mse_slots, n->mse_post, mse_d, n->L, ctx->cur, nullptr, n->grad + n->grad_n, gscale);
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "mse_finish", e);
        return AEFFT_OK;
    }
    if (fused_upd) return fail(ctx, AEFFT_ESTATE, "apply: fused update without the operator-form tail");      // (cannot happen: fused_upd implies op_state)
    // post-update MSE (fft_backproplib.cu:1460-1463): G = F.C of every eligible pair in one launch, then every pair's pass
    // over X with the MSE epilogue in one launch; pairs the fused form does not serve (dD == 1, B == 1) go pair by pair
    std::vector<char> g_in_S(n->L, 0);       // pair l: S holds G of the updated weights after this call
    {
        const bool nofuse = flag(AEFFT_F_NOFUSEMSE), nogroup = flag(AEFFT_F_NOGROUP);
        Contract gq[8], mq[8];
        int m = 0;
        std::vector<int> rest;
        for (int l = 0; l < n->L; ++l) {
            Pair& q = n->pr[l];
            if (!nofuse && !nogroup && q.dD >= 2 && n->B >= 2 && m < 8) {
                gq[m] = mk_G(q.F, q.C, q.G, q.dM, q.dD, q.P);
                mq[m] = mk_gmse(q.G, q.X, q.F, q.b, q.p, n->mse_slots + (size_t)l * MSE_SLOTS * MSE_SLOT_STRIDE, n->B, q.dM, q.dD, q.Nx, q.Ny);
                ++m;
            } else rest.push_back(l);
        }
        bool grouped = false;
        if (m > 1) {
            // G = F.C/(dM dD).  HBM-sized kernel spectra (no pooling): as the spectrum of the (2Nk-1)^2-tap kernel f (*) c
            // (weight_kernels.hip), which reads the kernels and writes dD*dD planes instead of reading all 2*dM*dD planes of C|F;
            // cache-sized ones: as a per-bin contraction of the spectra (measured faster there).
            const bool noq = flag(AEFFT_F_NOQPATH);
            double cf_bytes = 0;
            for (int l = 0; l < n->L; ++l) cf_bytes += 2.0 * n->pr[l].dM * n->pr[l].dD * n->pr[l].P * 8.0;
            bool gtaps = false;
            if (!noq && (cf_bytes > 256e6 || flag(AEFFT_F_GTAPS)) && m == n->L) RET_IF(gprime_from_taps(n, &gtaps));
            if (!gtaps) RET_IF(do_contract_group(ctx, gq, m, m, 0));
            ContractN g{};
            double bytes = 0;
            for (int i = 0; i < m; ++i) { g.q[i] = mq[i]; bytes += contract_bytes(mq[i]); }
            g.n = m;
            hipError_t e;
            {
                Bracket br(ctx, KID_CONTRACT, bytes);
                e = launch_contract_mfma(g, ctx->cur);
            }
            if (e == hipSuccess) grouped = true;
            else if (e != hipErrorInvalidValue) return fail(ctx, AEFFT_EHIP, "contract(mse group)", e);
            else (void)hipGetLastError();
        }
        for (int l = 0; l < n->L; ++l) {
            const bool in_group = grouped && std::find(rest.begin(), rest.end(), l) == rest.end();
            bool left = in_group;
            if (!in_group) RET_IF(reforward_mse(n, n->pr[l], n->mse_slots + (size_t)l * MSE_SLOTS * MSE_SLOT_STRIDE, &left));
            g_in_S[l] = left;
        }
    }
    {
        Bracket br(ctx, KID_DIFFMSE, 4.0 * n->L * MSE_SLOTS);
        Pair& ql = n->pr[n->L - 1];
        BetaArgs ba{ql.beta, ql.F, ql.b, ql.p, ql.dM, ql.dD, ql.P};
        const bool want_beta = g_in_S[n->L - 1] && ql.beta && ql.dD <= 256;
        hipError_t e = launch_mse_finish(n->mse_slots, n->mse_post, mse_d, n->L, ctx->cur, want_beta ? &ba : nullptr, n->grad + n->grad_n, gscale);     // also the copy-out to mse_d and to the packed buffer's tail
        ql.G_valid = want_beta && e == hipSuccess;
        if (e != hipSuccess) return fail(ctx, AEFFT_EHIP, "mse_finish", e);
    }
    return AEFFT_OK;
}

// input prefetch bookkeeping: everything of step k that reads this step's input-spectra buffer has been enqueued
static int mark_step_point(aefft_net* n)
{
    if (!n->input_ready || !n->ev_end[0]) return AEFFT_OK;
    HIPCHK(n->ctx, hipEventRecord(n->ev_end[n->step_no & 1], n->ctx->stream));
    n->ev_end_valid[n->step_no & 1] = true;
    return AEFFT_OK;
}

extern "C" int aefft_net_set_input_ready(aefft_net* n, int enable)
{
    if (!n) return AEFFT_EINVAL;
    aefft_ctx* ctx = n->ctx;
    if (enable && !n->X0alt) {
        const Pair& q = n->pr[0];
        RET_IF(net_alloc_t(n, &n->X0alt, (size_t)n->B * q.dD * q.P));
        for (int i = 0; i < 2; ++i) HIPCHK(ctx, hipEventCreateWithFlags(&n->ev_end[i], hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&n->ev_r2c, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&n->ev_mid, hipEventDisableTiming));
    }
    n->input_ready = enable != 0;
    return AEFFT_OK;
}

extern "C" int aefft_net_step_grad(aefft_net* n, const float* frames_d, float* recon_d)
{
    if (!n) return AEFFT_EINVAL;
    aefft_ctx* ctx = n->ctx;
    ++n->step_no;
    RET_IF(net_forward(n, frames_d, recon_d, true, op_eligible(n)));
    {
        int rcg = grads_grouped(n);
        if (rcg == AEFFT_OK) rcg = mse_flush(n);      // (a gradient route without the wgrad launch: the deferred MSE sums as their own launch after all)
        if (rcg != AEFFT_OK) { n->recon_deferred = nullptr; return rcg; }
    }
    if (n->recon_deferred) {
        // The reconstruction's inverse FFT starts HERE: where a data-parallel run waits for its all-reduce the GPU is otherwise
        // idle, and what follows on this stream (update, spectra, MSE) is latency-bound.  Joined by aefft_net_step_apply,
        // aefft_sync or the next call on this net.
        float* recon = n->recon_deferred;
        n->recon_deferred = nullptr;
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[0], ctx->ev_fork, 0));
        ctx->cur = ctx->aux[0];
        const int rc = launch_recon(n, recon, WS_MID3);
        ctx->cur = ctx->stream;
        RET_IF(rc);
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
        ctx->recon_join = true;
    }
    if (n->recon_pending) {
        // the documented default: recon_d is complete, in stream order on the context stream, when this call's work is
        // (include/aefft.h; the pipelined mode relaxes it).  Joining later -- behind the update half -- was measured: see DESIGN.md 6.
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[0], 0));
        n->recon_pending = false;
    }
    n->have_grad = true;
    if (n->input_ready && n->ev_mid) { HIPCHK(ctx, hipEventRecord(n->ev_mid, ctx->stream)); n->ev_mid_valid = true; }
    return mark_step_point(n);
}

// 8-bit frames: the same calls with the input transform converting on load (fft_kernels.hip r2c_rows_kernel<N, true>); nothing else reads the frames
extern "C" int aefft_net_step_grad_u8(aefft_net* n, const unsigned char* frames_d, float* recon_d)
{
    if (!n) return AEFFT_EINVAL;
    n->ctx->in_u8 = true;
    const int rc = aefft_net_step_grad(n, reinterpret_cast<const float*>(frames_d), recon_d);
    n->ctx->in_u8 = false;
    return rc;
}
extern "C" int aefft_net_forward_u8(aefft_net* n, const unsigned char* frames_d, float* recon_d)
{
    if (!n) return AEFFT_EINVAL;
    n->ctx->in_u8 = true;
    const int rc = aefft_net_forward(n, reinterpret_cast<const float*>(frames_d), recon_d);
    n->ctx->in_u8 = false;
    return rc;
}

extern "C" int aefft_net_step_form(aefft_net* n)
{
    if (!n) return -1;
    if (!op_eligible(n)) return AEFFT_FORM_PER_FRAME;
    const bool chain = n->Wp && (n->compact || n->L == 1) && chain_switches_ok();
    return chain ? AEFFT_FORM_OPERATOR_CHAIN : AEFFT_FORM_OPERATOR;
}

extern "C" int aefft_net_grad_buffer(aefft_net* n, float** buf_d, size_t* nfloats)
{
    if (!n) return AEFFT_EINVAL;
    if (buf_d) *buf_d = n->grad;
    if (nfloats) *nfloats = n->grad_n + (size_t)n->L;
    return AEFFT_OK;
}

extern "C" int aefft_net_last_mse(aefft_net* n, float* mse_d)
{
    if (!n || !mse_d) return AEFFT_EINVAL;
    aefft_ctx* ctx = n->ctx;
    RET_IF(mse_flush(n));
    HIPCHK(ctx, hipMemcpyAsync(mse_d, n->mse_post, sizeof(float) * n->L, hipMemcpyDeviceToDevice, ctx->stream));
    return AEFFT_OK;
}

extern "C" int aefft_net_step_apply(aefft_net* n, float del0, int maxdiff, int sym, float grad_scale, float* mse_d)
{
    if (!n) return AEFFT_EINVAL;
    aefft_ctx* ctx = n->ctx;
    if (!n->have_grad) return fail(ctx, AEFFT_ESTATE, "aefft_net_step_apply: call aefft_net_step_grad first");
    RET_IF(mse_flush(n));
    for (auto& q : n->pr) q.G_valid = false;          // the weights are about to change (the grouped path re-derives G and sets it again)
    const float del = 0.1f * del0;
    RET_IF(apply_grouped(n, del, maxdiff, sym, grad_scale, mse_d));
    n->have_grad = false;
    n->upd_after_fwd = true;
    RET_IF(join_recon(ctx));
    return mark_step_point(n);
}
