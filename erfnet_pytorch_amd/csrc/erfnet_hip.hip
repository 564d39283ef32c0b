// liberfnet_hip.so — C ABI (include/erfnet_hip.h) over the gfx950 kernels in igemm.h / aux_kernels.h.
// Host side only builds descriptors and launches; it never allocates, copies or synchronises.
#include "../../include/erfnet_hip.h"
#include "aux_kernels.h"
#include "igemm.h"
#include "stream16.h"
#include "sampler16.h"
#include "pair.h"
#include "line_bf16.h"
#include "line2_bf16.h"
#include "line_x3.h"
#include "line_x3r.h"
#include "igemm_bf16.h"
#include "igemm_x3.h"
#include "wgrad_line.h"
#include "wgrad_line_bf16.h"
#include "wgrad_mm.h"
#include "loss_kernels.h"
#include "metric_kernels.h"
#include "optim_kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace {

thread_local char g_err[512] = "";
std::atomic<unsigned long long> g_launches{0};

// Kernel-path switches: -1 = not read yet (the environment variable decides), else 0 / 1.
// erf_set_option overrides them at run time (A/B tests in one process).
struct Switch {
    std::atomic<int> v{-1};
    const char *off_env;        // the variable that flips the default
    int dflt;
    explicit Switch(const char *env, int default_on = 1) : off_env(env), dflt(default_on) {}
    bool on()
    {
        int x = v.load(std::memory_order_relaxed);
        if (x < 0) {
            const char *e = getenv(off_env);
            x = (e && atoi(e)) ? 1 - dflt : dflt;
            v.store(x, std::memory_order_relaxed);
        }
        return x != 0;
    }
    int set(int value)
    {
        const int prev = on() ? 1 : 0;
        v.store(value ? 1 : 0, std::memory_order_relaxed);
        return prev;
    }
};
Switch g_sw_pair("ERFNET_NO_PAIR"), g_sw_wgrad_line("ERFNET_NO_WGRAD_LINE"), g_sw_stream16("ERFNET_NO_STREAM16");
// off by default: measured 5-12 % slower than two conv16_kernel launches (both are issue-bound, not
// memory-bound, at 16 channels); ERFNET_PAIR16=1 or ERF_OPT_PAIR16 turns it on
Switch g_sw_pair16("ERFNET_PAIR16", 0);
Switch g_sw_conv16_rows("ERFNET_NO_CONV16_ROWS");      // ERF_OPT_CONV16_ROWS
// the split-bf16 convolutions on the rolling kernel (line_x3r_kernel: every input line split once
// instead of once per tap); ERFNET_NO_X3R=1 / ERF_OPT_X3_ROLL = 0 keeps line_x3_kernel
Switch g_sw_x3r("ERFNET_NO_X3R");
// DownsamplerBlock 16 -> 64 backward: the max-pool's gradient formed in the data-gradient kernel's
// epilogue; ERFNET_NO_POOL_FUSE=1 / ERF_OPT_POOL_FUSE = 0 keeps pool_bwd_kernel
Switch g_sw_pool_fuse("ERFNET_NO_POOL_FUSE");
std::atomic<int> g_line2{1};        // ERF_OPT_LINE2
// fp32 64 / 128-channel 1-D convolutions on the bf16 matrix pipe, operands split in three bf16 terms
// (line_x3_kernel, wgrad_line_x3_kernel): fp32 accuracy at 2.7x the fp32 matrix rate.
// Mode 1 (default): where it pays — a convolution launch takes the split kernel when it has at least
// two line tiles per resident block (below that the kernel's weight prologue outweighs its faster
// tiles and the fused fp32 pair wins); 2: always; 0 (ERFNET_NO_X3=1, ERF_OPT_SPLIT_BF16 = 0): never,
// i.e. the fp32 MFMA kernels (pair_f32_kernel, wgrad_line_kernel).
std::atomic<int> g_x3_mode{-1};
int x3_mode()
{
    int x = g_x3_mode.load(std::memory_order_relaxed);
    if (x < 0) {
        const char *e = getenv("ERFNET_NO_X3");
        x = (e && atoi(e)) ? 0 : 1;
        const char *f = getenv("ERFNET_X3_ALWAYS");
        if (x && f && atoi(f)) x = 2;
        g_x3_mode.store(x, std::memory_order_relaxed);
    }
    return x;
}

int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// Launch a kernel template instantiated for the storage type of the activations: inside the
// argument, T is float (dt == 0) or bf16_t (dt != 0).
#define BY_DT(dt, ...)                                                                             \
    do {                                                                                           \
        if (dt) { using T = bf16_t; __VA_ARGS__; }                                                 \
        else { using T = float; __VA_ARGS__; }                                                     \
    } while (0)

#define CHECK_LAUNCH(name)                                                                         \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return fail("%s launch failed: %s", name, hipGetErrorString(e_));    \
        g_launches.fetch_add(1, std::memory_order_relaxed);                                        \
    } while (0)

// Bump allocator over the caller's workspace (256-byte granules).
struct Arena {
    char *base;
    size_t size, off;
    Arena(void *p, size_t n) : base((char *)p), size(n), off(0) {}
    void *take(size_t bytes)
    {
        const size_t a = (off + 255) & ~(size_t)255;
        if (!base || a + bytes > size) return nullptr;
        off = a + bytes;
        return base + a;
    }
    float *floats(size_t n) { return (float *)take(n * sizeof(float)); }
};

const Epilogue kNoEpi = {nullptr, {nullptr, nullptr, nullptr, nullptr, 0.0f, 0, 0}, nullptr,
                         nullptr, 0, 0, nullptr, 0, 0, 0};

BnAffine bn_eval_affine(const erf_bn_t *bn)
{
    BnAffine a = {bn->gamma, bn->beta, bn->running_mean, bn->running_var, bn->eps, 0, 1};
    return a;
}
BnAffine bn_saved_affine(const erf_bn_t *bn)
{
    BnAffine a = {bn->gamma, bn->beta, bn->save_mean, bn->save_invstd, bn->eps, 1, 1};
    return a;
}
inline int bn_mode(const erf_bn_t *bn) { return bn ? bn->mode : 0; }

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// ------------------------------------------------------------------ bf16 line kernel (line_bf16.h)
int cu_count()
{
    static std::atomic<int> n{0};
    int v = n.load(std::memory_order_relaxed);
    if (v <= 0) {
        int dev = 0;
        v = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int c = 0;
            if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0) v = c;
        }
        n.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Block tile (pixels of a line) of the bf16 line kernel for NC channels and lines of L pixels;
// 0 = no kernel.  A pair needs the whole line in one tile; a single convolution may cut the line
// into segments.
int line_tile_bf16(int NC, int L, bool pair)
{
    if (NC == 128) {
        if (L <= 32) return 32;
        if (L <= 64) return 64;
        if (L <= 128) return 128;
        return pair ? 0 : 128;
    }
    if (NC == 64) {
        if (L <= 64) return 64;
        if (L <= 128) return 128;
        if (L <= 256) return 256;
        return pair ? 0 : 256;
    }
    return 0;
}

size_t line_packed_elems(int NC) { return (size_t)3 * NC * NC; }

void line_pack_desc(LinePackDesc &k, const float *w, bf16_t *wp, int NC, int wsk, int wsn, const int *woff,
                    bool x3 = false)
{
    memset(&k, 0, sizeof(k));
    k.w = w; k.wp = wp; k.NC = NC; k.wsk = wsk; k.wsn = wsn; k.x3 = x3 ? 1 : 0;
    for (int t = 0; t < 3; ++t) k.woff[t] = woff[t];
}

int launch_line_pack(const LinePackDesc *pks, int n, hipStream_t st)
{
    for (int first = 0; first < n; first += ERF_LINE_PACK_BATCH) {
        const int cnt = n - first < ERF_LINE_PACK_BATCH ? n - first : ERF_LINE_PACK_BATCH;
        LinePackBatch b;
        memset(&b, 0, sizeof(b));
        int most = 0;
        for (int i = 0; i < cnt; ++i) {
            b.p[i] = pks[first + i];
            most = b.p[i].NC > most ? b.p[i].NC : most;
        }
        if (!most) continue;
        line_pack_kernel<<<dim3((unsigned)(3 * most * most / 256 / 4), cnt, 1), 256, 0, st>>>(b);
        CHECK_LAUNCH("line_pack_kernel");
    }
    return 0;
}

// Pairs whose blocks see at least two tiles take the eight-wave kernel (conv_a of tile k+1 beside
// conv_b of tile k); ERFNET_NO_LINE2=1 keeps them on the four-wave one (experiments, A/B tests).
bool line2_takes(const LineDesc &ld, int NC, int BM, bool pair)
{
    static int no_line2 = -1;
    if (no_line2 < 0) { const char *e = getenv("ERFNET_NO_LINE2"); no_line2 = (e && atoi(e)) ? 1 : 0; }
    const int ntiles = ld.nlines * ld.segs;
    const int cus = cu_count();
    const int grid = ntiles < cus ? ntiles : cus;
    if (!pair || no_line2 || !g_line2.load(std::memory_order_relaxed) || ld.segs != 1 || ntiles < 2 * grid) return false;
    return (NC == 128 && (BM == 32 || BM == 64)) || (NC == 64 && (BM == 64 || BM == 128));
}
int run_line(const LineDesc &ld, int NC, int BM, bool pair, hipStream_t st)
{
    const int ntiles = ld.nlines * ld.segs;
    if (ntiles <= 0) return 0;
    const int cus = cu_count();
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    if (line2_takes(ld, NC, BM, pair)) {
#define LINE2_CASE(nc, bm)                                                                          \
        if (NC == nc && BM == bm) {                                                                 \
            line2_bf16_kernel<nc, bm><<<grid, 512, 0, st>>>(ld);                                    \
            CHECK_LAUNCH("line2_bf16_kernel");                                                      \
            return 0;                                                                               \
        }
        LINE2_CASE(128, 32) LINE2_CASE(128, 64) LINE2_CASE(64, 64) LINE2_CASE(64, 128)
#undef LINE2_CASE
    }
    if (ld.epi1.bits_out || ld.epi1.mask_bits || ld.epi2.bits_out || ld.epi2.mask_bits)
        return fail("bf16 line kernel: mask bits belong to the eight-wave pair kernel only");
#define LINE_CASE(nc, bm)                                                                           \
    if (NC == nc && BM == bm) {                                                                     \
        if (pair) line_bf16_kernel<nc, bm, true><<<grid, 256, 0, st>>>(ld);                        \
        else line_bf16_kernel<nc, bm, false><<<grid, 256, 0, st>>>(ld);                            \
        CHECK_LAUNCH("line_bf16_kernel");                                                           \
        return 0;                                                                                   \
    }
    LINE_CASE(128, 32) LINE_CASE(128, 64) LINE_CASE(128, 128)
    LINE_CASE(64, 64) LINE_CASE(64, 128) LINE_CASE(64, 256)
#undef LINE_CASE
    return fail("bf16 line kernel: no variant for %d channels, tile %d", NC, BM);
}

// A 3-tap 1-D convolution described as an implicit GEMM -> geometry of the line kernel.  The taps
// run across the lines: a convolution along W takes image columns as lines, one along H image rows.
// Returns false when the convolution does not fit the kernel.
bool line_geometry(const IgemmDesc &d, LineDesc &ld, bool reads_memory = true)
{
    const int NC = d.K;
    if (NC != 64 && NC != 128) return false;
    if (d.Nout != NC || d.in_C != NC || d.out_C != NC || d.in_coff || d.out_coff) return false;
    if (d.ss != 1 || d.os != 1 || d.ngroups != 1 || d.gtap0[0] != 0 || d.gntaps[0] != 3) return false;
    if (d.goa[0] || d.gob[0] || d.epi.nscale) return false;
    if (d.in_H != d.GH || d.in_W != d.GW || d.out_H != d.GH || d.out_W != d.GW) return false;
    // (the second convolution of a pair takes its input from LDS: its `in` may alias anything)
    if (!aligned16(d.in) || !aligned16(d.out) || (reads_memory && d.in == d.out)) return false;
    const long long P = (long long)d.N * d.GH * d.GW;
    if (P * NC >= (1LL << 31)) return false;
    if (d.epi.res && (d.epi.res_C != NC || d.epi.res_coff || !aligned16(d.epi.res))) return false;
    if (d.epi.mask && (d.epi.mask_C != NC || d.epi.mask_coff || !aligned16(d.epi.mask))) return false;
    if (d.tdh[1] || d.tdw[1]) return false;                 // the centre tap stays on the line
    bool along_w = true, along_h = true;
    for (int t = 0; t < 3; ++t) { along_w = along_w && d.tdh[t] == 0; along_h = along_h && d.tdw[t] == 0; }
    if (!along_w && !along_h) return false;
    memset(&ld, 0, sizeof(ld));
    ld.in = (const bf16_t *)d.in; ld.out = (bf16_t *)d.out;
    ld.epi1 = d.epi; ld.epi2 = kNoEpi;
    ld.N = d.N; ld.img_pixels = d.GH * d.GW;
    if (along_w) {
        ld.lines_per_img = d.GW; ld.L = d.GH; ld.pstride = d.GW; ld.qstride = 1;
        for (int t = 0; t < 3; ++t) ld.s1[t] = d.tdw[t];
    } else {
        ld.lines_per_img = d.GH; ld.L = d.GW; ld.pstride = 1; ld.qstride = d.GW;
        for (int t = 0; t < 3; ++t) ld.s1[t] = d.tdh[t];
    }
    ld.nlines = d.N * ld.lines_per_img;
    return true;
}
const int kCfgLine = 101;
// the same geometry on fp32 tensors (line_x3_kernel): byte offsets must stay below 2^31
const int kCfgLineX3 = 102;
int line_x3_tile(int NC, int L) { return (NC == 128 && L <= 32) ? 32 : 64; }
int line_x3_slots(int NC) { return cu_count() * (NC == 64 ? 2 : 1); }
bool line_x3_ok(const IgemmDesc &d, LineDesc &ld)
{
    const int mode = x3_mode();
    if (d.dt || !mode) return false;
    if ((long long)d.N * d.GH * d.GW * d.K >= (1LL << 29)) return false;
    if (!line_geometry(d, ld)) return false;
    if (mode >= 2) return true;
    const int BM = line_x3_tile(d.K, ld.L);
    const long long ntiles = (long long)ld.nlines * ((ld.L + BM - 1) / BM);
    return ntiles >= 2LL * line_x3_slots(d.K);
}
// blocks line_x3_kernel is launched with for this convolution (rows of IgemmDesc::bn_partial)
int line_x3_grid(const IgemmDesc &d)
{
    LineDesc ld;
    if (!line_geometry(d, ld)) return 0;
    const int BM = line_x3_tile(d.K, ld.L);
    const long long ntiles = (long long)ld.nlines * ((ld.L + BM - 1) / BM);
    const int slots = line_x3_slots(d.K);
    return (int)(ntiles < slots ? ntiles : slots);
}
// Whether a launch of this geometry goes to the rolling kernel (line_x3r_kernel): 64-pixel tiles,
// taps {-dil, 0, +dil} in either order, whole tap chains per image, no output statistics.
// ERFNET_X3R_ONLY=64 / 128 keeps it to that width (experiments).
bool line_x3r_takes(const LineDesc &ld, int NC)
{
    static int only = -1;
    if (only < 0) { const char *e = getenv("ERFNET_X3R_ONLY"); only = e ? atoi(e) : 0; }
    const int dil = ld.s1[0] < 0 ? -ld.s1[0] : ld.s1[0];
    return g_sw_x3r.on() && !ld.stats && line_x3_tile(NC, ld.L) == 64 && dil > 0 && ld.s1[1] == 0 &&
           ld.s1[2] == -ld.s1[0] && ld.lines_per_img % dil == 0 && (!only || only == NC);
}
// The same question for a planned convolution (only the rolling kernel reads and writes mask bits).
bool igemm_on_x3r(const IgemmDesc &d)
{
    LineDesc ld;
    if (d.cfg != kCfgLineX3 || !line_geometry(d, ld)) return false;
    ld.stats = d.bn_partial;
    return line_x3r_takes(ld, d.K);
}
int run_line_x3(LineDesc ld, int NC, hipStream_t st)
{
    // 64 channels: two independent blocks per CU (each fills the other's barrier, latency and
    // epilogue gaps); 128 channels with lines of at most 32 pixels: 32-pixel tiles
    const int BM = line_x3_tile(NC, ld.L);
    ld.segs = (ld.L + BM - 1) / BM;
    const int ntiles = ld.nlines * ld.segs;
    if (ntiles <= 0) return 0;
    const int slots = line_x3_slots(NC);
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    if (line_x3r_takes(ld, NC)) {
        if (NC == 128) line_x3r_kernel<128, 64><<<grid, 256, 0, st>>>(ld);
        else line_x3r_kernel<64, 64><<<grid, 256, 0, st>>>(ld);
        CHECK_LAUNCH("line_x3r_kernel");
        return 0;
    }
    if (ld.epi1.bits_out || ld.epi1.mask_bits) return fail("line_x3_kernel: mask bits belong to the rolling kernel only");
#define X3_LAUNCH(nc, bm)                                                                           \
    do {                                                                                            \
        if (ld.stats) line_x3_kernel<nc, bm, true><<<grid, 256, 0, st>>>(ld);                       \
        else line_x3_kernel<nc, bm, false><<<grid, 256, 0, st>>>(ld);                               \
    } while (0)
    if (NC == 128 && BM == 32) X3_LAUNCH(128, 32);
    else if (NC == 128) X3_LAUNCH(128, 64);
    else if (ld.stats) return fail("line_x3_kernel: output statistics are built for 128 channels only");
    else line_x3_kernel<64, 64, false><<<grid, 256, 0, st>>>(ld);
#undef X3_LAUNCH
    CHECK_LAUNCH("line_x3_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------- igemm launch
// Tile configurations <WM, WN, GM, GN>: block tile BM = GM*WM*32 pixels x BN = GN*WN*32 channels.
struct TileCfg { int bm, bn; };
const TileCfg kCfg[] = {
    {128, 128}, {96, 128}, {64, 128}, {32, 128},     // 0-3: Nout > 64
    {128, 64}, {64, 64},                             // 4-5: 32 < Nout <= 64
    {256, 32}, {128, 32},                            // 6-7: Nout <= 32
    {128, 32},                                       // 8  : scalar-load variant
};

// Channels per K-step: 16 when the contraction has only 16 channels (no zero-padded MFMAs),
// otherwise 32 (fewer barriers per FLOP).  ERFNET_IGEMM_KC = 16 / 32 forces one (experiments).
int igemm_kc(int K)
{
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("ERFNET_IGEMM_KC");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 16 || forced == 32) return forced;
    return K <= 16 ? 16 : 32;
}

// Choose the tile: blocks are dealt round-robin over 256 CUs, so the kernel lasts as long as the
// CU with the most blocks: cost = ceil(blocks / 256) * BM.  Ties go to the larger tile (fewer
// weight-tile reloads per FLOP).
int pick_cfg(long long M, int first, int count)
{
    int best = first;
    long long best_cost = -1;
    for (int c = first; c < first + count; ++c) {
        const long long blocks = (M + kCfg[c].bm - 1) / kCfg[c].bm;
        const long long cost = ((blocks + 255) / 256) * kCfg[c].bm;
        if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
    return best;
}

// The 16-channel streaming kernel (conv16_kernel) takes stride-1, single-group, 3-tap convolutions
// whose tensors allow 16-byte accesses and 32-bit element offsets.  ERFNET_NO_STREAM16=1 turns the
// path off (experiments).
const int kCfgStream16 = 100;
bool stream16_enabled() { return g_sw_stream16.on(); }
bool stream16_ok(const IgemmDesc &d)
{
    if (!stream16_enabled() || !d.vec_ok) return false;
    if (d.K != 16 || d.Nout != 16 || d.ss != 1 || d.os != 1 || d.ngroups != 1 || d.GW < 16) return false;
    if (d.gtap0[0] != 0 || d.gntaps[0] != 3 || d.goa[0] || d.gob[0]) return false;
    if (d.out_C % 4 || d.out_coff % 4 || !aligned16(d.out)) return false;
    if (d.epi.res && (d.epi.res_C % 4 || d.epi.res_coff % 4 || !aligned16(d.epi.res))) return false;
    if (d.epi.mask && (d.epi.mask_C % 4 || d.epi.mask_coff % 4 || !aligned16(d.epi.mask))) return false;
    if (d.epi.bias && !aligned16(d.epi.bias)) return false;
    if (d.epi.nscale && !aligned16(d.epi.nscale)) return false;
    if (d.epi.bn.on && (!d.epi.bn.gamma || !d.epi.bn.beta || !aligned16(d.epi.bn.gamma) || !aligned16(d.epi.bn.beta) ||
                        !aligned16(d.epi.bn.mean) || !aligned16(d.epi.bn.var))) return false;
    const long long lim = 1LL << 29;    // buffer descriptors: at most 2^31 bytes
    const long long pin = (long long)d.N * d.in_H * d.in_W, pout = (long long)d.N * d.out_H * d.out_W;
    if (pin * d.in_C >= lim || pout * d.out_C >= lim) return false;
    if (d.epi.res && pout * d.epi.res_C >= lim) return false;
    if (d.epi.mask && pout * d.epi.mask_C >= lim) return false;
    return true;
}

// bf16 tensors on the bf16 MFMA (igemm_bf16_kernel): tiles {pixels, channels} and its requirements.
// ERFNET_NO_IGEMM_BF16=1 keeps bf16 tensors on the fp32 MFMA (experiments).
const int kCfgBf16 = 200;
const TileCfg kCfgB[] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}, {128, 32}};
bool igemm_bf16_ok(const IgemmDesc &d)
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("ERFNET_NO_IGEMM_BF16"); off = (e && atoi(e)) ? 1 : 0; }
    if (!d.dt || off) return false;
    if (d.K % 8 || d.in_C % 8 || d.in_coff % 8 || !aligned16(d.in)) return false;
    if (d.Nout % 4 || d.out_C % 4 || d.out_coff % 4 || ((uintptr_t)d.out & 7)) return false;
    if (d.epi.res && (d.epi.res_C % 4 || d.epi.res_coff % 4 || ((uintptr_t)d.epi.res & 7))) return false;
    if (d.epi.mask && (d.epi.mask_C % 4 || d.epi.mask_coff % 4 || ((uintptr_t)d.epi.mask & 7))) return false;
    if (d.epi.nscale && !aligned16(d.epi.nscale)) return false;
    return true;
}

// fp32 tensors with split operands on the bf16 MFMA (igemm_x3_kernel), tiles {pixels, 64 channels}.
// Follows the split-arithmetic switch of the line kernels; ERFNET_NO_IGEMM_X3=1 keeps these shapes on the
// fp32 MFMA (experiments).
const int kCfgX3 = 300;
bool igemm_x3_ok(const IgemmDesc &d)
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("ERFNET_NO_IGEMM_X3"); off = (e && atoi(e)) ? 1 : 0; }
    if (d.dt || off || x3_mode() == 0) return false;
    // not the 3-tap 1-D convolutions of a non_bottleneck_1d: where the line kernels leave one of those to the
    // generic path, the fused fp32 pair may take its neighbour, and a run of blocks has to equal its blocks
    // called one by one bit for bit
    if (d.ss == 1 && d.os == 1 && d.ngroups == 1 && d.gntaps[0] == 3) return false;
    if (d.K % 8 || d.in_C % 4 || d.in_coff % 4 || !aligned16(d.in)) return false;
    if (d.Nout % 4 || d.out_C % 4 || d.out_coff % 4 || !aligned16(d.out)) return false;
    if (d.epi.res && (d.epi.res_C % 4 || d.epi.res_coff % 4 || !aligned16(d.epi.res))) return false;
    if (d.epi.mask && (d.epi.mask_C % 4 || d.epi.mask_coff % 4 || !aligned16(d.epi.mask))) return false;
    if (d.epi.nscale && !aligned16(d.epi.nscale)) return false;
    const long long lim = 1LL << 29;
    const long long pin = (long long)d.N * d.in_H * d.in_W, pout = (long long)d.N * d.out_H * d.out_W;
    if (pin * d.in_C >= lim || pout * d.out_C >= lim) return false;
    if (d.epi.res && pout * d.epi.res_C >= lim) return false;
    if (d.epi.mask && pout * d.epi.mask_C >= lim) return false;
    return true;
}

// Validate, choose the tile, reserve the packed-weight buffer and describe the packing job.
int plan_igemm(IgemmDesc &d, Arena &ar, PackDesc &pk)
{
    if (d.K <= 0 || d.Nout <= 0) return fail("igemm: empty contraction");
    const long long M = (long long)d.N * d.GH * d.GW;
    if (d.in_H >= 32768 || d.in_W >= 32768 || d.GH * d.ss >= 65536 || d.GW * d.ss >= 32768)
        return fail("igemm: spatial extent too large");
    if (M * d.os * d.os >= (1LL << 31) || (long long)d.N * d.in_H * d.in_W >= (1LL << 31))
        return fail("igemm: more than 2^31 pixels");
    {   // buffer descriptors address at most 2^31 bytes per tensor
        const long long lim = 1LL << 29, pin = (long long)d.N * d.in_H * d.in_W, pout = (long long)d.N * d.out_H * d.out_W;
        if (pin * d.in_C >= lim || pout * d.out_C >= lim || (d.epi.res && pout * d.epi.res_C >= lim) ||
            (d.epi.mask && pout * d.epi.mask_C >= lim))
            return fail("igemm: tensors of 2 GiB or more are not supported");
    }
    d.vec_ok = (d.in_C % 4 == 0 && d.in_coff % 4 == 0 && d.K % 4 == 0 && aligned16(d.in)) ? 1 : 0;
    {
        LineDesc ld;
        if (line_x3_ok(d, ld)) {     // fp32 on the bf16 matrix pipe: three weight planes
            bf16_t *wp = (bf16_t *)ar.take(3 * line_packed_elems(d.K) * sizeof(bf16_t));
            if (!wp) return fail("igemm: workspace too small for the weight fragments");
            d.cfg = kCfgLineX3;
            d.Npad = d.Nout;
            d.wp = wp;
            memset(&pk, 0, sizeof(pk));
            pk.w = d.w; pk.K = d.K; pk.Nout = d.Nout; pk.wsk = d.wsk; pk.wsn = d.wsn;
            for (int t = 0; t < 3; ++t) pk.woff[t] = d.woff[t];
            pk.line = 2; pk.lwp = wp;
            return 0;
        }
    }
    if (d.dt) {          // bf16: the 64 / 128-channel 1-D convolutions take the line kernel
        LineDesc ld;
        if (line_geometry(d, ld)) {
            bf16_t *wp = (bf16_t *)ar.take(line_packed_elems(d.K) * sizeof(bf16_t));
            if (!wp) return fail("igemm: workspace too small for the weight fragments");
            d.cfg = kCfgLine;
            d.Npad = d.Nout;
            d.wp = wp;
            memset(&pk, 0, sizeof(pk));
            pk.w = d.w; pk.K = d.K; pk.Nout = d.Nout; pk.wsk = d.wsk; pk.wsn = d.wsn;
            for (int t = 0; t < 3; ++t) pk.woff[t] = d.woff[t];
            pk.line = 1; pk.lwp = wp;
            return 0;
        }
    }
    if (stream16_ok(d)) {          // 16 -> 16 channels: LDS-free streaming kernel, unpacked weights
        d.cfg = kCfgStream16;
        d.Npad = 16;
        memset(&pk, 0, sizeof(pk));
        return 0;
    }
    if (igemm_x3_ok(d)) {           // fp32 tensors, split operands on the bf16 MFMA: three packed bf16 planes
        const long long b128 = (M + 127) / 128, b64 = (M + 63) / 64;
        const long long c128 = ((b128 + 511) / 512) * 128, c64 = ((b64 + 511) / 512) * 64;   // two blocks per CU
        d.cfg = kCfgX3 + (c64 < c128 ? 1 : 0);
        d.Npad = (d.Nout + 63) / 64 * 64;
        int ntaps = 0;
        for (int g = 0; g < d.ngroups; ++g) ntaps = d.gtap0[g] + d.gntaps[g] > ntaps ? d.gtap0[g] + d.gntaps[g] : ntaps;
        const int kchunks = (d.K + 31) / 32;
        const size_t n = (size_t)ntaps * kchunks * d.Npad * 32;
        void *wp = ar.take(3 * n * sizeof(bf16_t));
        if (!wp) return fail("igemm: workspace too small for the split packed weights (%zu bf16)", 3 * n);
        d.wp = wp;
        d.x3_plane = (int)n;
        memset(&pk, 0, sizeof(pk));
        pk.w = d.w; pk.wp = (float *)wp; pk.bf16 = 3;
        pk.ntaps = ntaps; pk.K = d.K; pk.Nout = d.Nout; pk.Npad = d.Npad; pk.wsk = d.wsk; pk.wsn = d.wsn;
        pk.kc = 32;
        for (int t = 0; t < ERF_MAX_TAPS; ++t) pk.woff[t] = d.woff[t];
        return 0;
    }
    if (igemm_bf16_ok(d)) {         // bf16 tensors: bf16 MFMA, bf16 packed weights
        const int first = d.Nout > 64 ? 0 : (d.Nout > 32 ? 2 : 4), count = d.Nout > 32 ? 2 : 1;
        int best = first;
        long long best_cost = -1;
        for (int c = first; c < first + count; ++c) {
            const long long blocks = (M + kCfgB[c].bm - 1) / kCfgB[c].bm;
            const long long cost = ((blocks + 255) / 256) * kCfgB[c].bm;
            if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
        }
        d.cfg = kCfgBf16 + best;
        const int bn = kCfgB[best].bn;
        d.Npad = (d.Nout + bn - 1) / bn * bn;
        int ntaps = 0;
        for (int g = 0; g < d.ngroups; ++g) ntaps = d.gtap0[g] + d.gntaps[g] > ntaps ? d.gtap0[g] + d.gntaps[g] : ntaps;
        const int kchunks = (d.K + 31) / 32;
        const size_t n = (size_t)ntaps * kchunks * d.Npad * 32;
        void *wp = ar.take(n * sizeof(bf16_t));
        if (!wp) return fail("igemm: workspace too small for the packed weights (%zu bf16)", n);
        d.wp = wp;
        memset(&pk, 0, sizeof(pk));
        pk.w = d.w; pk.wp = (float *)wp; pk.bf16 = 1;
        pk.ntaps = ntaps; pk.K = d.K; pk.Nout = d.Nout; pk.Npad = d.Npad; pk.wsk = d.wsk; pk.wsn = d.wsn;
        pk.kc = 32;
        for (int t = 0; t < ERF_MAX_TAPS; ++t) pk.woff[t] = d.woff[t];
        return 0;
    }
    if (!d.vec_ok) d.cfg = 8;
    else if (d.Nout > 64) d.cfg = pick_cfg(M, 0, 4);
    else if (d.Nout > 32) d.cfg = pick_cfg(M, 4, 2);
    else d.cfg = pick_cfg(M, 6, 2);
    const int bn = kCfg[d.cfg].bn;
    d.Npad = (d.Nout + bn - 1) / bn * bn;
    int ntaps = 0;
    for (int g = 0; g < d.ngroups; ++g) ntaps = d.gtap0[g] + d.gntaps[g] > ntaps ? d.gtap0[g] + d.gntaps[g] : ntaps;
    const int KC = igemm_kc(d.K);
    const int kchunks = (d.K + KC - 1) / KC;
    const size_t n = (size_t)ntaps * kchunks * d.Npad * KC;
    float *wp = ar.floats(n);
    if (!wp) return fail("igemm: workspace too small for the packed weights (%zu floats)", n);
    d.wp = wp;
    memset(&pk, 0, sizeof(pk));
    pk.w = d.w; pk.wp = wp;
    pk.ntaps = ntaps; pk.K = d.K; pk.Nout = d.Nout; pk.Npad = d.Npad; pk.wsk = d.wsk; pk.wsn = d.wsn;
    pk.kc = KC;
    for (int t = 0; t < ERF_MAX_TAPS; ++t) pk.woff[t] = d.woff[t];
    return 0;
}

int launch_pack(const PackDesc *pks, int n, hipStream_t st)
{
    {   // fragments of the bf16 line kernel
        LinePackDesc lp[ERF_LINE_PACK_BATCH];
        int nl = 0;
        for (int i = 0; i < n; ++i) {
            if (!pks[i].line) continue;
            line_pack_desc(lp[nl++], pks[i].w, pks[i].lwp, pks[i].K, pks[i].wsk, pks[i].wsn, pks[i].woff, pks[i].line == 2);
            if (nl == ERF_LINE_PACK_BATCH) { if (launch_line_pack(lp, nl, st)) return 1; nl = 0; }
        }
        if (nl && launch_line_pack(lp, nl, st)) return 1;
    }
    for (int first = 0; first < n; first += ERF_PACK_BATCH) {
        const int cnt = n - first < ERF_PACK_BATCH ? n - first : ERF_PACK_BATCH;
        PackBatch b;
        memset(&b, 0, sizeof(b));
        size_t most = 0;
        for (int i = 0; i < cnt; ++i) {
            b.p[i] = pks[first + i];
            if (!b.p[i].kc) continue;          // convolution that reads torch's layout directly
            const size_t tot = (size_t)b.p[i].ntaps * ((b.p[i].K + b.p[i].kc - 1) / b.p[i].kc) * b.p[i].Npad * b.p[i].kc;
            most = tot > most ? tot : most;
        }
        if (most == 0) continue;
        dim3 grid((unsigned)((most + 255) / 256), cnt, 1);
        if (grid.x > 512) grid.x = 512;
        pack_weights_kernel<<<grid, 256, 0, st>>>(b);
        CHECK_LAUNCH("pack_weights_kernel");
    }
    return 0;
}

template <typename T>
int run_igemm_t(const IgemmDesc &d, hipStream_t st)
{
    const long long M = (long long)d.N * d.GH * d.GW;
    if (M == 0) return 0;
    if (d.cfg == kCfgLine) {
        LineDesc ld;
        if (!line_geometry(d, ld)) return fail("igemm: line plan without line geometry");
        ld.wp1 = (const bf16_t *)d.wp;
        const int BM = line_tile_bf16(d.K, ld.L, false);
        ld.segs = (ld.L + BM - 1) / BM;
        return run_line(ld, d.K, BM, false, st);
    }
    if (d.cfg == kCfgLineX3) {
        LineDesc ld;
        if (!line_geometry(d, ld)) return fail("igemm: line plan without line geometry");
        ld.wp1 = (const bf16_t *)d.wp;
        ld.stats = d.bn_partial;
        return run_line_x3(ld, d.K, st);
    }
    if (d.cfg == kCfgStream16) {
        // tile groups per wave: the per-wave set-up (descriptors, weights, coordinates) is paid once,
        // as long as about 12 waves per CU remain (ERFNET_CONV16_G overrides, experiments)
        static int forced = -1;
        if (forced < 0) { const char *e = getenv("ERFNET_CONV16_G"); forced = e ? atoi(e) : 0; }
        const int grp = forced ? forced : (M >= 3072LL * 256 ? 4 : (M >= 3072LL * 128 ? 2 : 1));
        if constexpr (std::is_same<T, bf16_t>::value) {
            // bf16 MFMA variant (16-byte operand loads): the weights are rounded to bf16 as in every other bf16 kernel
            static int no16 = -1;
            if (no16 < 0) { const char *e = getenv("ERFNET_NO_CONV16_BF16"); no16 = (e && atoi(e)) ? 1 : 0; }
            if (!no16 && d.in_C % 8 == 0 && d.in_coff % 8 == 0 && aligned16(d.in)) {
                // rows variant: tiles of 16 pixels never leave an image row, taps along one axis only
                const bool norows = !g_sw_conv16_rows.on();
                const bool same = d.in_H == d.GH && d.in_W == d.GW && d.out_H == d.GH && d.out_W == d.GW;
                const bool aw = d.tdh[0] == 0 && d.tdh[1] == 0 && d.tdh[2] == 0;
                const bool ah = d.tdw[0] == 0 && d.tdw[1] == 0 && d.tdw[2] == 0;
                const long long widest = M * (long long)std::max(std::max(d.in_C, d.out_C), std::max(d.epi.res ? d.epi.res_C : 0, d.epi.mask ? d.epi.mask_C : 0)) * 2;
                if (!norows && same && (aw || ah) && d.GW % 16 == 0 && widest < (1LL << 31) &&
                    d.out_C % 4 == 0 && d.out_coff % 4 == 0 && aligned16(d.out) &&
                    (!d.epi.res || (d.epi.res_C % 4 == 0 && d.epi.res_coff % 4 == 0 && aligned16(d.epi.res))) &&
                    (!d.epi.mask || (d.epi.mask_C % 4 == 0 && d.epi.mask_coff % 4 == 0 && aligned16(d.epi.mask)))) {
                    if (grp >= 2) {
                        const unsigned blocks = (unsigned)((M + 1023) / 1024);
                        if (aw) conv16_bf16_rows_kernel<8, 2, true><<<blocks, 256, 0, st>>>(d);
                        else conv16_bf16_rows_kernel<8, 2, false><<<blocks, 256, 0, st>>>(d);
                    } else {
                        const unsigned blocks = (unsigned)((M + 511) / 512);
                        if (aw) conv16_bf16_rows_kernel<8, 1, true><<<blocks, 256, 0, st>>>(d);
                        else conv16_bf16_rows_kernel<8, 1, false><<<blocks, 256, 0, st>>>(d);
                    }
                    CHECK_LAUNCH("conv16_bf16_rows_kernel");
                    return 0;
                }
                if (grp >= 2) conv16_bf16_kernel<8, 2><<<(unsigned)((M + 1023) / 1024), 256, 0, st>>>(d);
                else conv16_bf16_kernel<8, 1><<<(unsigned)((M + 511) / 512), 256, 0, st>>>(d);
                CHECK_LAUNCH("conv16_bf16_kernel");
                return 0;
            }
        }
        if constexpr (std::is_same<T, float>::value) {
            // rows variant (fp32): the same conditions as the bf16 one; bit-identical to conv16_kernel
            const bool same = d.in_H == d.GH && d.in_W == d.GW && d.out_H == d.GH && d.out_W == d.GW;
            const bool aw = d.tdh[0] == 0 && d.tdh[1] == 0 && d.tdh[2] == 0;
            const bool ah = d.tdw[0] == 0 && d.tdw[1] == 0 && d.tdw[2] == 0;
            if (g_sw_conv16_rows.on() && same && (aw || ah) && d.GW % 16 == 0) {    // (stream16_ok: 16-byte tensors, < 2^29 elements)
#define ROWS_CASE(u, gg)                                                                            \
                do {                                                                                \
                    const unsigned blocks = (unsigned)((M + 64 * u * gg - 1) / (64 * u * gg));      \
                    if (aw) conv16_rows_kernel<u, gg, true><<<blocks, 256, 0, st>>>(d);             \
                    else conv16_rows_kernel<u, gg, false><<<blocks, 256, 0, st>>>(d);               \
                } while (0)
                if (grp >= 4) ROWS_CASE(4, 4);
                else if (grp >= 2) ROWS_CASE(4, 2);
                else ROWS_CASE(4, 1);
#undef ROWS_CASE
                CHECK_LAUNCH("conv16_rows_kernel");
                return 0;
            }
        }
        if (grp >= 4) conv16_kernel<T, 3, 4, 4><<<(unsigned)((M + 1023) / 1024), 256, 0, st>>>(d);
        else if (grp >= 2) conv16_kernel<T, 3, 4, 2><<<(unsigned)((M + 511) / 512), 256, 0, st>>>(d);
        else conv16_kernel<T, 3, 4><<<(unsigned)((M + 255) / 256), 256, 0, st>>>(d);
        CHECK_LAUNCH("conv16_kernel");
        return 0;
    }
    if (d.cfg >= kCfgX3) {
        if constexpr (std::is_same<T, float>::value) {
            const int bm = d.cfg == kCfgX3 ? 128 : 64;
            dim3 grid((unsigned)((M + bm - 1) / bm), d.Npad / 64, d.ngroups);
            if (bm == 128) igemm_x3_kernel<2, 1, 2, 2, 2><<<grid, 256, 0, st>>>(d);
            else igemm_x3_kernel<1, 1, 2, 2, 2><<<grid, 256, 0, st>>>(d);
            CHECK_LAUNCH("igemm_x3_kernel");
            return 0;
        }
        return fail("igemm: split-operand plan on a bf16 tensor");
    }
    if (d.cfg >= kCfgBf16) {
        const TileCfg c = kCfgB[d.cfg - kCfgBf16];
        dim3 grid((unsigned)((M + c.bm - 1) / c.bm), d.Npad / c.bn, d.ngroups);
        switch (d.cfg - kCfgBf16) {
        case 0: igemm_bf16_kernel<2, 2, 2, 2, 2><<<grid, 256, 0, st>>>(d); break;
        case 1: igemm_bf16_kernel<1, 2, 2, 2, 2><<<grid, 256, 0, st>>>(d); break;
        case 2: igemm_bf16_kernel<2, 1, 2, 2, 2><<<grid, 256, 0, st>>>(d); break;
        case 3: igemm_bf16_kernel<1, 1, 2, 2, 2><<<grid, 256, 0, st>>>(d); break;
        default: igemm_bf16_kernel<1, 1, 4, 1, 2><<<grid, 256, 0, st>>>(d); break;
        }
        CHECK_LAUNCH("igemm_bf16_kernel");
        return 0;
    }
    const TileCfg c = kCfg[d.cfg];
    dim3 grid((unsigned)((M + c.bm - 1) / c.bm), d.Npad / c.bn, d.ngroups);
#define IG_LAUNCH(KC, MW, MW64)                                                                     \
    switch (d.cfg) {                                                                               \
    case 0: igemm_f32_kernel<2, 2, 2, 2, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    case 1: igemm_f32_kernel<3, 1, 1, 4, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    case 2: igemm_f32_kernel<2, 1, 1, 4, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    case 3: igemm_f32_kernel<1, 1, 1, 4, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    case 4: igemm_f32_kernel<1, 2, 4, 1, KC, MW64, true, T><<<grid, 256, 0, st>>>(d); break;         \
    case 5: igemm_f32_kernel<1, 1, 2, 2, KC, MW64, true, T><<<grid, 256, 0, st>>>(d); break;         \
    case 6: igemm_f32_kernel<2, 1, 4, 1, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    case 7: igemm_f32_kernel<1, 1, 4, 1, KC, MW, true, T><<<grid, 256, 0, st>>>(d); break;           \
    default: igemm_f32_kernel<1, 1, 4, 1, KC, MW, false, T><<<grid, 256, 0, st>>>(d); break;         \
    }
    // minimum blocks per CU (register budget): 64-column tiles keep three blocks resident
    if (igemm_kc(d.K) == 16) { IG_LAUNCH(16, 3, 3) } else { IG_LAUNCH(32, 2, 3) }
#undef IG_LAUNCH
    CHECK_LAUNCH("igemm_f32_kernel");
    return 0;
}
int run_igemm(const IgemmDesc &d, hipStream_t st)
{
    return d.dt ? run_igemm_t<bf16_t>(d, st) : run_igemm_t<float>(d, st);
}

// Single convolution: pack its weights, then run it.
int launch_igemm(IgemmDesc &d, Arena &ar, hipStream_t st)
{
    PackDesc pk;
    if (plan_igemm(d, ar, pk)) return 1;
    if (launch_pack(&pk, 1, st)) return 1;
    return run_igemm(d, st);
}

// ------------------------------------------------------------------ fused convolution pairs
// Two consecutive 1-D convolutions of a non_bottleneck_1d (the first across the lines the second
// runs along) go into one launch of pair_f32_kernel when the line fits a block tile.
// ERFNET_NO_PAIR=1 keeps every convolution in a launch of its own (experiments, A/B tests).
bool pair_enabled() { return g_sw_pair.on(); }

// Tile variant (block pixels) for a line of L pixels and NC channels, 0 = none.
int pair_tile(int NC, int L)
{
    if (NC == 128) return L <= 16 ? 0 : (L <= 32 ? 32 : (L <= 64 ? 64 : (L <= 128 ? 128 : 0)));
    if (NC == 64) return L <= 32 ? 0 : (L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 256 ? 256 : 0)));
    return 0;
}

// Two chained 16-channel convolutions (pair16_kernel): each as conv16_kernel takes it, one along
// each image axis, the second reaching at most 4 pixels.
const int kPair16 = 16;
bool pair16_side_ok(const IgemmDesc &d, const IgemmDesc &a)
{
    if (d.K != 16 || d.Nout != 16 || d.ss != 1 || d.os != 1 || d.ngroups != 1 || d.GW < 16) return false;
    if (d.gtap0[0] != 0 || d.gntaps[0] != 3 || d.goa[0] || d.gob[0]) return false;
    if (d.N != a.N || d.GH != a.GH || d.GW != a.GW || d.in_H != a.GH || d.in_W != a.GW || d.out_H != a.GH ||
        d.out_W != a.GW) return false;
    if (d.in_C % 4 || d.in_coff % 4 || !aligned16(d.in)) return false;
    if (d.out_C % 4 || d.out_coff % 4 || !aligned16(d.out)) return false;
    if (d.epi.res && (d.epi.res_C % 4 || d.epi.res_coff % 4 || !aligned16(d.epi.res))) return false;
    if (d.epi.mask && (d.epi.mask_C % 4 || d.epi.mask_coff % 4 || !aligned16(d.epi.mask))) return false;
    if (d.epi.bias && !aligned16(d.epi.bias)) return false;
    if (d.epi.nscale && !aligned16(d.epi.nscale)) return false;
    if (d.epi.bn.on && (!d.epi.bn.gamma || !d.epi.bn.beta || !aligned16(d.epi.bn.gamma) || !aligned16(d.epi.bn.beta) ||
                        !aligned16(d.epi.bn.mean) || !aligned16(d.epi.bn.var))) return false;
    const long long lim = 1LL << 29, px = (long long)d.N * d.GH * d.GW;
    if (px * d.in_C >= lim || px * d.out_C >= lim) return false;
    if (d.epi.res && px * d.epi.res_C >= lim) return false;
    if (d.epi.mask && px * d.epi.mask_C >= lim) return false;
    return true;
}
// 0 = no; else fills the axis / reach of the second convolution
bool pair16_ok(const IgemmDesc &a, const IgemmDesc &b, int *b_along_w = nullptr, int *halo = nullptr)
{
    if (!stream16_enabled() || !g_sw_pair16.on() || a.dt != b.dt) return false;
    if (!pair16_side_ok(a, a) || !pair16_side_ok(b, a)) return false;
    if (b.in != a.out || a.out_C != 16 || a.out_coff || b.in_C != 16 || b.in_coff) return false;
    bool a_w = true, a_h = true, b_w = true, b_h = true;
    int reach = 0;
    for (int t = 0; t < 3; ++t) {
        a_w = a_w && a.tdh[t] == 0; a_h = a_h && a.tdw[t] == 0;
        b_w = b_w && b.tdh[t] == 0; b_h = b_h && b.tdw[t] == 0;
        const int r = abs(b.tdh[t]) + abs(b.tdw[t]);
        reach = r > reach ? r : reach;
    }
    if (!((a_w && b_h) || (a_h && b_w)) || reach < 1 || reach > 4) return false;
    if (b_along_w) *b_along_w = (a_h && b_w) ? 1 : 0;
    if (halo) *halo = reach;
    return true;
}

// a = first convolution, b = second (b.in is a.out).  Returns the tile variant or 0.
int pair_variant(const IgemmDesc &a, const IgemmDesc &b, bool ignore_switch = false)
{
    if (!ignore_switch && !pair_enabled()) return 0;
    const int NC = a.K;
    if (a.dt != b.dt) return 0;
    if (pair16_ok(a, b)) return kPair16;
    if (a.dt) {          // bf16: both halves must fit the line kernel, the second running along the first's lines
        LineDesc la, lb;
        if (!line_geometry(a, la) || !line_geometry(b, lb, false) || b.in != a.out || b.K != NC) return 0;
        if (a.N != b.N || a.GH != b.GH || a.GW != b.GW) return 0;
        if (a.epi.res && b.epi.res) return 0;               // one residual tile per launch
        if (la.pstride == lb.pstride) return 0;             // same axis: not a pair
        return line_tile_bf16(NC, la.L, true);
    }
    if (!ignore_switch) {       // each half takes line_x3_kernel where that is faster than the fp32 pair
        LineDesc la, lb;
        if (line_x3_ok(a, la) && line_x3_ok(b, lb)) return 0;
    }
    const IgemmDesc *dd[2] = {&a, &b};
    for (const IgemmDesc *d : dd) {
        if (d->K != NC || d->Nout != NC || d->in_C != NC || d->out_C != NC || d->in_coff || d->out_coff) return 0;
        if (d->ss != 1 || d->os != 1 || d->ngroups != 1 || d->gtap0[0] != 0 || d->gntaps[0] != 3) return 0;
        if (d->goa[0] || d->gob[0] || d->epi.nscale) return 0;
        if (d->N != a.N || d->GH != a.GH || d->GW != a.GW || d->in_H != a.GH || d->in_W != a.GW ||
            d->out_H != a.GH || d->out_W != a.GW) return 0;
        if (!aligned16(d->in) || !aligned16(d->out)) return 0;
        const long long P = (long long)d->N * d->GH * d->GW;
        if (P * NC >= (1LL << 29)) return 0;
        if (d->epi.res && P * d->epi.res_C >= (1LL << 29)) return 0;
        if (d->epi.mask && P * d->epi.mask_C >= (1LL << 29)) return 0;
    }
    if (b.in != a.out) return 0;
    bool a_w = true, a_h = true, b_w = true, b_h = true;
    for (int t = 0; t < 3; ++t) {
        a_w = a_w && a.tdh[t] == 0; a_h = a_h && a.tdw[t] == 0;
        b_w = b_w && b.tdh[t] == 0; b_h = b_h && b.tdw[t] == 0;
    }
    if (a_w && b_h) return pair_tile(NC, a.GH);        // lines = image columns
    if (a_h && b_w) return pair_tile(NC, a.GW);        // lines = image rows
    return 0;
}

// Reserve and describe the packed weights (16 channels per K-step) and fill the kernel descriptor.
// keep_mid = false drops the store of the intermediate tensor.
// A fused pair as planned: the fp32 kernel's descriptor, or the bf16 line kernel's.
struct FusedPair {
    int dt;
    int p16_on;             // 1 = the 16-channel kernel (either storage type)
    PairDesc pd;
    LineDesc ld;
    Pair16Desc p16;
};

int plan_pair(const IgemmDesc &a, const IgemmDesc &b, bool keep_mid, Arena &ar, PackDesc pk[2], FusedPair &fp)
{
    const int NC = a.K;
    const IgemmDesc *dd[2] = {&a, &b};
    fp.dt = a.dt;
    if (b.out == a.in || (keep_mid && a.out == a.in))
        return fail("conv pair: the output must not alias the input (other blocks still read it)");
    fp.p16_on = 0;
    {
        int along_w = 0, halo = 0;
        if (pair16_ok(a, b, &along_w, &halo)) {
            Pair16Desc &q = fp.p16;
            q.a = a; q.b = b;
            q.keep_mid = keep_mid ? 1 : 0;
            q.b_along_w = along_w; q.halo = halo;
            q.th = along_w ? 8 : 16; q.tw = along_w ? 64 : 32;
            q.tiles_h = (a.GH + q.th - 1) / q.th; q.tiles_w = (a.GW + q.tw - 1) / q.tw;
            memset(&pk[0], 0, sizeof(PackDesc));        // torch's weight layout is read directly
            memset(&pk[1], 0, sizeof(PackDesc));
            fp.p16_on = 1;
            return 0;
        }
    }
    if (a.dt) {
        LineDesc &ld = fp.ld, lb;
        if (!line_geometry(a, ld) || !line_geometry(b, lb, false)) return fail("conv pair (bf16): geometry");
        bf16_t *wp = (bf16_t *)ar.take(2 * line_packed_elems(NC) * sizeof(bf16_t));
        if (!wp) return fail("conv pair: workspace too small for the weight fragments");
        for (int i = 0; i < 2; ++i) {
            PackDesc &k = pk[i];
            memset(&k, 0, sizeof(k));
            k.w = dd[i]->w; k.K = NC; k.Nout = NC; k.wsk = dd[i]->wsk; k.wsn = dd[i]->wsn;
            for (int t = 0; t < 3; ++t) k.woff[t] = dd[i]->woff[t];
            k.line = 1; k.lwp = wp + i * line_packed_elems(NC);
        }
        ld.wp1 = wp; ld.wp2 = wp + line_packed_elems(NC);
        ld.mid = keep_mid ? (bf16_t *)a.out : nullptr;
        ld.out = (bf16_t *)b.out;
        ld.epi2 = b.epi;
        ld.segs = 1;
        // the second convolution runs along the line: its tap offsets in line positions
        for (int t = 0; t < 3; ++t) ld.s2[t] = ld.pstride == 1 ? b.tdw[t] : b.tdh[t];
        return 0;
    }
    PairDesc &pd = fp.pd;
    const float *wp[2];
    for (int i = 0; i < 2; ++i) {
        const size_t n = (size_t)3 * NC * NC;
        float *p = ar.floats(n);
        if (!p) return fail("conv pair: workspace too small for the packed weights (%zu floats)", n);
        wp[i] = p;
        PackDesc &k = pk[i];
        memset(&k, 0, sizeof(k));
        k.w = dd[i]->w; k.wp = p; k.ntaps = 3; k.K = NC; k.Nout = NC; k.Npad = NC;
        k.wsk = dd[i]->wsk; k.wsn = dd[i]->wsn; k.kc = 16;
        for (int t = 0; t < 3; ++t) k.woff[t] = dd[i]->woff[t];
    }
    memset(&pd, 0, sizeof(pd));
    pd.in = (const float *)a.in; pd.wp1 = wp[0]; pd.wp2 = wp[1];
    pd.mid = keep_mid ? (float *)a.out : nullptr;
    pd.out = (float *)b.out;
    pd.epi1 = a.epi; pd.epi2 = b.epi;
    pd.N = a.N; pd.img_pixels = a.GH * a.GW;
    const bool columns = a.tdh[0] == 0 && a.tdh[2] == 0;
    if (columns) {
        pd.lines_per_img = a.GW; pd.L = a.GH; pd.pstride = a.GW; pd.qstride = 1;
        for (int t = 0; t < 3; ++t) { pd.s1[t] = a.tdw[t]; pd.s2[t] = b.tdh[t]; }
    } else {
        pd.lines_per_img = a.GH; pd.L = a.GW; pd.pstride = 1; pd.qstride = a.GW;
        for (int t = 0; t < 3; ++t) { pd.s1[t] = a.tdh[t]; pd.s2[t] = b.tdw[t]; }
    }
    pd.nlines = a.N * pd.lines_per_img;
    return 0;
}

int run_pair_f32(const PairDesc &pd, int NC, int tile, hipStream_t st);
int run_pair(const FusedPair &fp, int NC, int tile, hipStream_t st)
{
    if (fp.p16_on) {
        const Pair16Desc &q = fp.p16;
        const long long blocks = (long long)q.a.N * q.tiles_h * q.tiles_w;
        if (blocks == 0) return 0;
        if (blocks >= (1LL << 31)) return fail("conv pair (16 channels): too many tiles");
        BY_DT(fp.dt, (pair16_kernel<T><<<(unsigned)blocks, 256, 0, st>>>(q)));
        CHECK_LAUNCH("pair16_kernel");
        return 0;
    }
    if (fp.dt) return run_line(fp.ld, NC, tile, true, st);
    return run_pair_f32(fp.pd, NC, tile, st);
}
int run_pair_f32(const PairDesc &pd, int NC, int tile, hipStream_t st)
{
    if (pd.nlines == 0) return 0;
    // fewer lines than two per CU: eight waves per block halve a block's own latency
    static int force_nw = -1;
    if (force_nw < 0) { const char *e = getenv("ERFNET_PAIR_WAVES"); force_nw = e ? atoi(e) : 0; }
    const bool wide = force_nw == 8 || (force_nw != 4 && pd.nlines <= 512);
    // fewer lines than half the CUs: split the second convolution's output channels over two blocks
    static int no_split = -1;
    if (no_split < 0) { const char *e = getenv("ERFNET_PAIR_NOSPLIT"); no_split = (e && atoi(e)) ? 1 : 0; }
    if (NC == 128 && tile == 64 && pd.nlines <= 128 && !no_split && force_nw == 0)
        pair_f32_kernel<64, 128, 4, 2><<<dim3(pd.nlines, 2, 1), 256, 0, st>>>(pd);
    else if (NC == 128 && tile == 64 && wide) pair_f32_kernel<64, 128, 8><<<pd.nlines, 512, 0, st>>>(pd);
    else if (NC == 64 && tile == 128 && wide) pair_f32_kernel<128, 64, 8><<<pd.nlines, 512, 0, st>>>(pd);
    // the large tiles always take eight waves (four would need more registers than a wave may hold)
    else if (NC == 128 && tile == 128 && force_nw != 4) pair_f32_kernel<128, 128, 8><<<pd.nlines, 512, 0, st>>>(pd);
    else if (NC == 64 && tile == 256 && force_nw != 4) pair_f32_kernel<256, 64, 8><<<pd.nlines, 512, 0, st>>>(pd);
    else if (NC == 128 && tile == 32) pair_f32_kernel<32, 128, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else if (NC == 64 && tile == 64) pair_f32_kernel<64, 64, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else if (NC == 128 && tile == 64) pair_f32_kernel<64, 128, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else if (NC == 128 && tile == 128) pair_f32_kernel<128, 128, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else if (NC == 64 && tile == 128) pair_f32_kernel<128, 64, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else if (NC == 64 && tile == 256) pair_f32_kernel<256, 64, 4><<<pd.nlines, 256, 0, st>>>(pd);
    else return fail("conv pair: no kernel for %d channels, tile %d", NC, tile);
    CHECK_LAUNCH("pair_f32_kernel");
    return 0;
}

// Weight gradients of up to four convolutions of identical geometry (the four 1-D convolutions of
// a non_bottleneck_1d) in one launch (blockIdx.z = convolution) followed by one slab reduction
// (blockIdx.y = convolution): one launch ramp and one tail instead of four.  `in`/`g`/taps may
// differ per convolution; K, Nout, ntaps, the pixel grid and the strides must agree.
// `bias_taps`: bit t set = the bias row of tap t enters db (taps whose g pixels partition the
// gradient tensor).
// A run of non_bottleneck_1d blocks in backward (erf_batch_begin / erf_batch_pack / erf_batch_end): the
// caller walks its blocks twice, each with a workspace of its own.  Pass 1 (collect) plans every block and
// keeps its weight-packing descriptors; erf_batch_pack packs them all in one launch; pass 2 (run) plans
// again (same workspace, same addresses), skips the packing and defers the slab reductions, which
// erf_batch_end launches together.  Per thread: a run's backward is one autograd node on one thread.
enum { BATCH_OFF = 0, BATCH_COLLECT = 1, BATCH_RUN = 2 };
struct BatchCtx {
    int mode = BATCH_OFF;
    std::vector<PackDesc> packs;
    struct Red { ReduceDesc r; long long nw; bool vec; };
    std::vector<Red> reds;
};
static thread_local BatchCtx g_batch;

int launch_reduce_batch(const WgradDesc *d, int n, float *const *dw, float *const *db, const int *woff,
                        int wsk, int wsn, unsigned bias_taps, hipStream_t st)
{
    ReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    for (int i = 0; i < n; ++i) {
        ReduceDesc &r = rb.r[i];
        r.partial = d[i].partial; r.dw = dw[i]; r.db = db ? db[i] : nullptr;
        r.nsplit = d[i].nsplit; r.ntaps = d[i].ntaps; r.K = d[i].K; r.Nout = d[i].Nout;
        for (int t = 0; t < d[i].ntaps; ++t) r.woff[t] = woff[t];
        r.wsk = wsk; r.wsn = wsn; r.bias_taps = bias_taps;
    }
    const long long nw = (long long)d[0].ntaps * d[0].K * d[0].Nout;
    const long long tot = nw + d[0].Nout;
    // 16-byte slab reads where every slab of the batch allows them (ERFNET_NO_REDUCE4=1: scalar kernel)
    static int no4 = -1;
    if (no4 < 0) { const char *e = getenv("ERFNET_NO_REDUCE4"); no4 = (e && atoi(e)) ? 1 : 0; }
    bool vec = !no4 && d[0].Nout % 4 == 0;
    for (int i = 0; i < n && vec; ++i)
        vec = d[i].K == d[0].K && d[i].Nout == d[0].Nout && d[i].ntaps == d[0].ntaps && aligned16(d[i].partial) &&
              ((long long)d[i].ntaps * d[i].K * d[i].Nout + (long long)d[i].ntaps * d[i].Nout) % 4 == 0;
    if (g_batch.mode == BATCH_RUN) {        // deferred: erf_batch_end launches it (the slabs stay where they are)
        for (int i = 0; i < n; ++i) g_batch.reds.push_back(BatchCtx::Red{rb.r[i], nw, vec});
        return 0;
    }
    if (vec) {
        const int wblocks = (int)((nw + 127) / 128), bblocks = (d[0].Nout + 31) / 32;
        reduce_partials4_kernel<<<dim3((unsigned)(wblocks + bblocks), n, 1), 256, 0, st>>>(rb, wblocks);
        CHECK_LAUNCH("reduce_partials4_kernel");
        return 0;
    }
    reduce_partials_kernel<<<dim3((unsigned)((tot + 31) / 32), n, 1), 256, 0, st>>>(rb);
    CHECK_LAUNCH("reduce_partials_kernel");
    return 0;
}

// The 3-tap 1-D convolutions at 64 / 128 channels take wgrad_line_kernel (all taps per block, X and
// G read once).  ERFNET_NO_WGRAD_LINE=1 keeps them on wgrad_f32_kernel (experiments).
bool wgrad_line_ok(const WgradDesc *dd, int n)
{
    if (!g_sw_wgrad_line.on()) return false;
    const int NC = dd[0].K;
    if (NC != 64 && NC != 128) return false;
    for (int i = 0; i < n; ++i) {
        const WgradDesc &e = dd[i];
        if (e.K != NC || e.Nout != NC || e.in_C != NC || e.g_C != NC || e.in_coff || e.g_coff) return false;
        if (e.ntaps != 3 || e.ss != 1 || e.os != 1) return false;
        if (e.in_H != e.GH || e.in_W != e.GW || e.g_H != e.GH || e.g_W != e.GW) return false;
        bool along_h = true, along_w = true;
        for (int t = 0; t < 3; ++t) {
            if (e.toa[t] || e.tob[t]) return false;
            along_h = along_h && e.tdw[t] == 0;
            along_w = along_w && e.tdh[t] == 0;
        }
        const int *td = along_h ? e.tdh : e.tdw;
        if (!(along_h || along_w) || td[1] != 0 || td[0] != -td[2] || td[2] < 1 || td[2] > 16) return false;
        if ((long long)e.N * e.GH * e.GW * NC >= (1LL << 29)) return false;
        // a line is cut into 64-position segments: short lines would leave most MFMA rows empty
        const int len = along_h ? e.GH : e.GW;
        if (len * 10 < ((len + 63) / 64) * 64 * 7) return false;
    }
    return true;
}

int launch_wgrad_line(WgradDesc *dd, int n, float *const *dw, float *const *db, const int *woff, int wsk,
                      int wsn, unsigned bias_taps, Arena &ar, hipStream_t st)
{
    const int NC = dd[0].K;
    const long long M = (long long)dd[0].N * dd[0].GH * dd[0].GW;
    if (M == 0) return 0;
    // 128 channels: one 512-thread block per CU owns all 128 x 128 x 3 outputs (X is staged once);
    // 64 channels: two 256-thread blocks per CU (slots below)
    const int jtiles = 1;
    // fp32 tensors on the bf16 matrix pipe (split operands): one block per CU; at 64 channels the
    // block's two wave groups write a slab each
    const bool x3 = !dd[0].dt && x3_mode() != 0;
    // 64 channels, dilation <= 4 (every 64-channel block of the network): a 72-row X tile, two
    // 256-thread blocks per CU (independent blocks fill each other's staging phases)
    int maxdil = 0;
    for (int i = 0; i < n; ++i) {
        const int dl = dd[i].tdw[0] == 0 && dd[i].tdw[2] == 0 ? dd[i].tdh[2] : dd[i].tdw[2];
        maxdil = dl > maxdil ? dl : maxdil;
    }
    static int no_small = -1;
    if (no_small < 0) { const char *e = getenv("ERFNET_X3_WGRAD_XR96"); no_small = (e && atoi(e)) ? 1 : 0; }
    const bool x3_small = x3 && NC == 64 && maxdil <= 4 && !no_small;
    const int kgroups = x3 && NC == 64 && !x3_small ? 2 : 1;
    // (64 channels, bf16 / fp32-MFMA kernels: two blocks per CU — a third one gains the kernel 0.6 us and
    // costs the slab reduction a third more slabs to read, 5.7 us per run of blocks)
    const int slots = x3_small ? 512 : ((NC == 128 || x3) ? 256 : 512);
    const size_t slab = (size_t)3 * NC * NC + 3 * NC;
    WgradLineBatch wb;
    memset(&wb, 0, sizeof(wb));
    int nsplit = slots / (jtiles * n);
    if (nsplit < 1) nsplit = 1;
    for (int i = 0; i < n; ++i) {
        WgradDesc &e = dd[i];
        WgradLineDesc &q = wb.d[i];
        const bool along_h = e.tdw[0] == 0 && e.tdw[2] == 0;
        q.in = e.in; q.g = e.g;
        q.N = e.N; q.img_pixels = e.GH * e.GW;
        if (along_h) { q.Len = e.GH; q.lines_per_img = e.GW; q.pstride = e.GW; q.qstride = 1; q.dil = e.tdh[2]; }
        else { q.Len = e.GW; q.lines_per_img = e.GH; q.pstride = 1; q.qstride = e.GW; q.dil = e.tdw[2]; }
        q.segs_per_line = (q.Len + 63) / 64;
        q.nseg = e.N * q.lines_per_img * q.segs_per_line;
        if (nsplit > q.nseg) nsplit = q.nseg;
    }
    // every convolution of the batch gets the same number of slabs; a block whose segment range
    // is empty writes a slab of zeros
    for (int i = 0; i < n; ++i) {
        WgradDesc &e = dd[i];
        WgradLineDesc &q = wb.d[i];
        q.spb = (q.nseg + nsplit - 1) / nsplit;
        e.nsplit = nsplit * kgroups;
        e.partial = ar.floats(slab * e.nsplit);
        if (!e.partial) return fail("wgrad: workspace too small (%zu floats needed)", slab * e.nsplit * n);
        q.partial = e.partial;
    }
    const int max_split = nsplit;
    dim3 grid((unsigned)max_split, jtiles, n);
    if (x3) {
        // 128 channels: four waves with 64 x 64 outputs each (192 accumulator registers, one wave per
        // SIMD) — the eight-wave 32 x 64 tiling reads 15 fragments per 36 MFMAs and is bound by LDS reads
        // (measured 122 us against 111); ERFNET_X3_WGRAD_8W=1 selects it (experiments)
        static int w8 = -1;
        if (w8 < 0) { const char *e = getenv("ERFNET_X3_WGRAD_8W"); w8 = (e && atoi(e)) ? 1 : 0; }
        if (NC == 128 && !w8) wgrad_line_x3_kernel<128, 2, 2, 2, 2, 1, 1><<<grid, 256, 0, st>>>(wb);
        else if (NC == 128) wgrad_line_x3_kernel<128, 1, 2, 4, 2, 1, 1><<<grid, 512, 0, st>>>(wb);
        else if (x3_small) wgrad_line_x3_kernel<64, 1, 1, 2, 2, 1, 2, 72><<<grid, 256, 0, st>>>(wb);
        else wgrad_line_x3_kernel<64, 1, 1, 2, 2, 2, 1><<<grid, 512, 0, st>>>(wb);
    } else if (dd[0].dt) {         // bf16 tensors: bf16 MFMA, fragments through the transposing LDS read
        if (NC == 128) wgrad_line_bf16_kernel<128, 1, 2, 4, 2, 1><<<grid, 512, 0, st>>>(wb);
        else wgrad_line_bf16_kernel<64, 1, 1, 2, 2, 3><<<grid, 256, 0, st>>>(wb);
    } else if (NC == 128) wgrad_line_kernel<128, 1, 2, 4, 2, 1, float><<<grid, 512, 0, st>>>(wb);
    else wgrad_line_kernel<64, 1, 1, 2, 2, 3, float><<<grid, 256, 0, st>>>(wb);
    CHECK_LAUNCH("wgrad_line_kernel");
    // the slab carries the column sums of G in bias row 0 only
    return launch_reduce_batch(dd, n, dw, db, woff, wsk, wsn, bias_taps ? 1u : 0u, st);
}

int launch_wgrad_batch(WgradDesc *dd, int n, float *const *dw, float *const *db, const int *woff, int wsk,
                       int wsn, unsigned bias_taps, Arena &ar, hipStream_t st)
{
    if (n < 1 || n > ERF_WGRAD_BATCH) return fail("wgrad: batch of %d", n);
    WgradDesc &d = dd[0];
    const long long M = (long long)d.N * d.GH * d.GW;
    if (M >= (1LL << 31)) return fail("wgrad: more than 2^31 pixels");
    bool vec = true, s16 = stream16_enabled() && d.K == 16 && d.Nout == 16 && d.ntaps == 3 && d.ss == 1 &&
                          d.os == 1 && d.in_H == d.GH && d.in_W == d.GW && d.g_H == d.GH && d.g_W == d.GW &&
                          d.GW >= 16;
    for (int i = 0; i < n; ++i) {
        const WgradDesc &e = dd[i];
        if (e.K != d.K || e.Nout != d.Nout || e.ntaps != d.ntaps || e.N != d.N || e.GH != d.GH || e.GW != d.GW)
            return fail("wgrad: batched convolutions differ in geometry");
        if ((long long)e.N * e.in_H * e.in_W * e.in_C >= (1LL << 29) || (long long)e.N * e.g_H * e.g_W * e.g_C >= (1LL << 29))
            return fail("wgrad: tensors of 2 GiB or more are not supported");
        vec = vec && e.in_C % 4 == 0 && e.in_coff % 4 == 0 && e.K % 4 == 0 && e.g_C % 4 == 0 &&
              e.g_coff % 4 == 0 && e.Nout % 4 == 0 && aligned16(e.in) && aligned16(e.g);
        s16 = s16 && !e.toa[0] && !e.tob[0] && !e.toa[1] && !e.tob[1] && !e.toa[2] && !e.tob[2];
    }
    if (s16) {      // 16 -> 16 channels: streaming kernel, one launch per convolution
        constexpr int NW = 8;
        const size_t slab = 3 * 256 + 3 * 16;
        long long nsplit = (M + NW * 64 - 1) / (NW * 64);          // >= 4 MFMA groups per wave
        if (nsplit > 512) nsplit = 512;
        if (nsplit < 1) nsplit = 1;
        // bf16 tensors whose channel slices allow 16-byte loads: bf16 MFMA, 32 pixels per instruction
        static int no16 = -1;
        if (no16 < 0) { const char *e = getenv("ERFNET_NO_WGRAD16_BF16"); no16 = (e && atoi(e)) ? 1 : 0; }
        bool b16 = d.dt != 0 && !no16;
        for (int i = 0; i < n; ++i)
            b16 = b16 && dd[i].in_C % 8 == 0 && dd[i].in_coff % 8 == 0 && dd[i].g_C % 8 == 0 && dd[i].g_coff % 8 == 0 &&
                  aligned16(dd[i].in) && aligned16(dd[i].g);
        // fp32 tensors: the same kernel with split operands (ERF_OPT_SPLIT_BF16); ERFNET_NO_WGRAD16_X3=1 keeps
        // them on the fp32 MFMA (experiments)
        static int no16x3 = -1;
        if (no16x3 < 0) { const char *e = getenv("ERFNET_NO_WGRAD16_X3"); no16x3 = (e && atoi(e)) ? 1 : 0; }
        bool x16 = d.dt == 0 && x3_mode() != 0 && !no16x3;
        for (int i = 0; i < n; ++i)
            x16 = x16 && dd[i].in_C % 4 == 0 && dd[i].in_coff % 4 == 0 && dd[i].g_C % 4 == 0 && dd[i].g_coff % 4 == 0 &&
                  aligned16(dd[i].in) && aligned16(dd[i].g);
        const int gran = NW * ((b16 || x16) ? 32 : 16);
        long long sp = (M + nsplit - 1) / nsplit;
        sp = (sp + gran - 1) / gran * gran;
        nsplit = (M + sp - 1) / sp;
        for (int i = 0; i < n; ++i) {
            WgradDesc &e = dd[i];
            e.nsplit = (int)nsplit;
            e.split_pixels = (int)sp;
            e.partial = ar.floats(slab * nsplit);
            if (!e.partial) return fail("wgrad: workspace too small (%zu floats needed)", slab * nsplit);
            if (b16) wgrad16_bf16_kernel<NW><<<(unsigned)nsplit, NW * 64, 0, st>>>(e);
            else if (x16) wgrad16_x3_kernel<NW><<<(unsigned)nsplit, NW * 64, 0, st>>>(e);
            else BY_DT(d.dt, (wgrad16_kernel<T, 3, NW><<<(unsigned)nsplit, NW * 64, 0, st>>>(e)));
            CHECK_LAUNCH("wgrad16_kernel");
        }
        return launch_reduce_batch(dd, n, dw, db, woff, wsk, wsn, bias_taps, st);
    }
    if (vec && wgrad_line_ok(dd, n)) return launch_wgrad_line(dd, n, dw, db, woff, wsk, wsn, bias_taps, ar, st);
    int IBLK, JBLK, cfg, PM = 32;
    if (!vec) { IBLK = 32; JBLK = 32; cfg = 7; PM = 64; }
    else if (d.K > 64 && d.Nout > 64) { IBLK = 128; JBLK = 128; cfg = 0; }
    else if (d.K > 64 && d.Nout > 32) { IBLK = 128; JBLK = 64; cfg = 1; }
    else if (d.K > 32 && d.Nout > 64) { IBLK = 64; JBLK = 128; cfg = 2; }
    else if (d.K > 32 && d.Nout > 32) { IBLK = 64; JBLK = 64; cfg = 3; }
    else if (d.K <= 32 && d.Nout > 32) { IBLK = 32; JBLK = 64; cfg = 4; PM = 128; }
    else if (d.K > 32) { IBLK = 64; JBLK = 32; cfg = 5; PM = 128; }
    else { IBLK = 32; JBLK = 32; cfg = 6; PM = 64; }
    const int itiles = (d.K + IBLK - 1) / IBLK, jtiles = (d.Nout + JBLK - 1) / JBLK;
    const int tiles = d.ntaps * itiles * jtiles;
    const size_t slab = (size_t)d.ntaps * d.K * d.Nout + (size_t)d.ntaps * d.Nout;
    // Pixels are split in slabs of `spb` PM-pixel steps.  Blocks are dealt round-robin over the
    // 256 CUs and co-resident blocks share the MFMA pipe, so the kernel lasts as long as the CU
    // with the most blocks; every slab costs one write and one read of the partial tile.
    const long long steps = (M + PM - 1) / PM;
    const double t_step = (IBLK / 64.0) * (JBLK / 64.0) * 0.43 * (PM / 32);   // us of one block step on a shared pipe
    const double slab_us = slab * sizeof(float) * 2 / 4.0e6;         // write + read at ~4 TB/s
    static int forced_spb = -1;
    if (forced_spb < 0) { const char *e = getenv("ERFNET_WGRAD_SPB"); forced_spb = e ? atoi(e) : 0; }
    long long best_spb = steps;
    double best = 1e30;
    for (long long spb = 2; spb <= steps; ++spb) {
        const long long ns = (steps + spb - 1) / spb;
        if (ns * (long long)(slab * sizeof(float)) * n > (96LL << 20)) continue;
        const long long per_cu = ((long long)n * tiles * ns + 255) / 256;
        double t = per_cu * spb * t_step * (per_cu == 1 ? 1.25 : 1.0) + n * ns * slab_us;
        if (t < best) { best = t; best_spb = spb; }
    }
    if (forced_spb > 0) best_spb = forced_spb < steps ? forced_spb : steps;
    const long long sp = best_spb * PM;
    const long long nsplit = (M + sp - 1) / sp;
    WgradBatch wb;
    memset(&wb, 0, sizeof(wb));
    for (int i = 0; i < n; ++i) {
        WgradDesc &e = dd[i];
        e.vec_ok = vec ? 1 : 0;
        e.itiles = itiles; e.jtiles = jtiles;
        e.nsplit = (int)nsplit;
        e.split_pixels = (int)sp;
        e.partial = ar.floats(slab * nsplit);
        if (!e.partial) return fail("wgrad: workspace too small (%zu floats needed)", slab * nsplit * n);
        wb.d[i] = e;
    }
    dim3 grid((unsigned)nsplit, tiles, n);
    // bf16 tensors, tiles of 64 / 128 channels: bf16 MFMA; ERFNET_NO_WGRAD_MM=1 keeps them on the fp32
    // MFMA (experiments)
    static int no_mm = -1;
    if (no_mm < 0) { const char *e = getenv("ERFNET_NO_WGRAD_MM"); no_mm = (e && atoi(e)) ? 1 : 0; }
    if (cfg <= 3 && !no_mm && d.dt) {
        if (cfg == 0) wgrad_mm_kernel<2, 2, 2, 2><<<grid, 256, 0, st>>>(wb);
        else if (cfg == 1) wgrad_mm_kernel<2, 1, 2, 2><<<grid, 256, 0, st>>>(wb);
        else if (cfg == 2) wgrad_mm_kernel<1, 2, 2, 2><<<grid, 256, 0, st>>>(wb);
        else wgrad_mm_kernel<1, 1, 2, 2><<<grid, 256, 0, st>>>(wb);
        CHECK_LAUNCH("wgrad_mm_kernel");
        return launch_reduce_batch(dd, n, dw, db, woff, wsk, wsn, bias_taps, st);
    }
    switch (cfg) {
    case 0: BY_DT(d.dt, (wgrad_f32_kernel<2, 2, 2, 2, true, 32, T><<<grid, 256, 0, st>>>(wb))); break;
    case 1: BY_DT(d.dt, (wgrad_f32_kernel<2, 1, 2, 2, true, 32, T><<<grid, 256, 0, st>>>(wb))); break;
    case 2: BY_DT(d.dt, (wgrad_f32_kernel<1, 2, 2, 2, true, 32, T><<<grid, 256, 0, st>>>(wb))); break;
    case 3: BY_DT(d.dt, (wgrad_f32_kernel<1, 1, 2, 2, true, 32, T><<<grid, 256, 0, st>>>(wb))); break;
    case 4: BY_DT(d.dt, (wgrad_f32_kernel<1, 1, 1, 2, true, 128, T><<<grid, 128, 0, st>>>(wb))); break;
    case 5: BY_DT(d.dt, (wgrad_f32_kernel<1, 1, 2, 1, true, 128, T><<<grid, 128, 0, st>>>(wb))); break;
    case 6: BY_DT(d.dt, (wgrad_f32_kernel<1, 1, 1, 1, true, 64, T><<<grid, 64, 0, st>>>(wb))); break;
    default: BY_DT(d.dt, (wgrad_f32_kernel<1, 1, 1, 1, false, 64, T><<<grid, 64, 0, st>>>(wb))); break;
    }
    CHECK_LAUNCH("wgrad_f32_kernel");
    return launch_reduce_batch(dd, n, dw, db, woff, wsk, wsn, bias_taps, st);
}

// Single convolution.
int launch_wgrad(WgradDesc &d, float *dw, float *db, const int *woff, int wsk, int wsn,
                 unsigned bias_taps, Arena &ar, hipStream_t st)
{
    float *dws[1] = {dw}, *dbs[1] = {db};
    return launch_wgrad_batch(&d, 1, dws, dbs, woff, wsk, wsn, bias_taps, ar, st);
}

// ------------------------------------------------------------------ 16-channel sampler layers
// sampler16.h pairs a coarse grid (K = 16, 32, 48 or 64 channels) with a fine grid of 16 channels.
bool sampler16_shape_ok(int K, long long coarse_pixels, int coarse_C, int fine_C)
{
    if (!stream16_enabled()) return false;
    if (K % 16 || K < 16 || K > 64) return false;
    return coarse_pixels * coarse_C < (1LL << 29) && coarse_pixels * 4 * fine_C < (1LL << 29);
}

// fine = epi(transposed 3x3 stride-2 convolution of coarse).  d: GH x GW = coarse grid, in = coarse,
// out = fine (16 channels at out_coff), w indexed woff = kh*3+kw, wsk (coarse channel), wsn (fine).
bool up16_fwd_ok(const IgemmDesc &d)
{
    if (d.Nout != 16 || d.epi.nscale) return false;
    if (!sampler16_shape_ok(d.K, (long long)d.N * d.GH * d.GW, d.in_C, d.out_C)) return false;
    if (d.in_C % 4 || d.in_coff % 4 || d.out_C % 4 || d.out_coff % 4 || !aligned16(d.in) || !aligned16(d.out)) return false;
    if (d.epi.res && (d.epi.res_C % 4 || d.epi.res_coff % 4 || !aligned16(d.epi.res) ||
                      (long long)d.N * d.GH * d.GW * 4 * d.epi.res_C >= (1LL << 29))) return false;
    if (d.epi.mask && (d.epi.mask_C % 4 || d.epi.mask_coff % 4 || !aligned16(d.epi.mask) ||
                       (long long)d.N * d.GH * d.GW * 4 * d.epi.mask_C >= (1LL << 29))) return false;
    if (d.epi.bias && !aligned16(d.epi.bias)) return false;
    if (d.epi.bn.on && (!d.epi.bn.gamma || !d.epi.bn.beta || !aligned16(d.epi.bn.gamma) || !aligned16(d.epi.bn.beta) ||
                        !aligned16(d.epi.bn.mean) || !aligned16(d.epi.bn.var))) return false;
    return true;
}
int launch_up16_fwd(const IgemmDesc &d, hipStream_t st)
{
    const long long M = (long long)d.N * d.GH * d.GW;
    if (M == 0) return 0;
    long long blocks = (M + 63) / 64;             // one 16-position tile per wave at least
    // about as many blocks as are resident at once (bf16: three per CU; fp32: two or three — 512 and 768
    // blocks measure the same): every block pays the weight fill, so further blocks cost more than they hide
    const long long cap = d.dt ? 768 : 512;
    if (blocks > cap) blocks = cap;
    if (d.dt && d.in_C % 8 == 0 && d.in_coff % 8 == 0 && d.K % 16 == 0 && d.K <= 64) {     // bf16 tensors: bf16 MFMA
        switch (d.K / 16) {
        case 1: up16_fwd_bf16_kernel<1><<<(unsigned)blocks, 256, 0, st>>>(d); break;
        case 2: up16_fwd_bf16_kernel<2><<<(unsigned)blocks, 256, 0, st>>>(d); break;
        case 3:
            if (d.pool_x) up16_fwd_bf16_kernel<3, true><<<(unsigned)blocks, 256, 0, st>>>(d);
            else up16_fwd_bf16_kernel<3><<<(unsigned)blocks, 256, 0, st>>>(d);
            break;
        default: up16_fwd_bf16_kernel<4><<<(unsigned)blocks, 256, 0, st>>>(d); break;
        }
        CHECK_LAUNCH("up16_fwd_bf16_kernel");
        return 0;
    }
    switch (d.K / 16) {
    case 1: BY_DT(d.dt, (up16_fwd_kernel<T, 1><<<(unsigned)blocks, 256, 0, st>>>(d))); break;
    case 2: BY_DT(d.dt, (up16_fwd_kernel<T, 2><<<(unsigned)blocks, 256, 0, st>>>(d))); break;
    case 3:
        if (d.pool_x) BY_DT(d.dt, (up16_fwd_kernel<T, 3, true><<<(unsigned)blocks, 256, 0, st>>>(d)));
        else BY_DT(d.dt, (up16_fwd_kernel<T, 3><<<(unsigned)blocks, 256, 0, st>>>(d)));
        break;
    default: BY_DT(d.dt, (up16_fwd_kernel<T, 4><<<(unsigned)blocks, 256, 0, st>>>(d))); break;
    }
    CHECK_LAUNCH("up16_fwd_kernel");
    return 0;
}

// Weight gradient of either layer.  d: GH x GW = coarse grid, in = coarse tensor (K channels at
// in_coff of in_C), g = fine tensor (16 channels at g_coff of g_C).  dw[woff = kh*3+kw + k*wsk + j*wsn];
// the bias gradient is the sum of the fine tensor (kside = 0) or of the coarse tensor (kside = 1).
int launch_up16_wgrad(WgradDesc &d, float *dw, float *db, int wsk, int wsn, int kside, Arena &ar, hipStream_t st)
{
    const long long M = (long long)d.N * d.GH * d.GW;
    const int KQ = d.K / 16;
    const size_t slab = (size_t)9 * d.K * 16 + 9 * 16 + d.K;
    long long nsplit = (M + 255) / 256;           // 64 positions per wave at least
    if (nsplit > 512) nsplit = 512;
    if (nsplit < 1) nsplit = 1;
    // bf16 tensors whose channel slices allow 16-byte loads: bf16 MFMA, 32 positions per instruction
    static int no16 = -1;
    if (no16 < 0) { const char *e = getenv("ERFNET_NO_WGRAD16_BF16"); no16 = (e && atoi(e)) ? 1 : 0; }
    const bool b16 = d.dt != 0 && !no16 && d.in_C % 8 == 0 && d.in_coff % 8 == 0 && d.g_C % 8 == 0 && d.g_coff % 8 == 0 &&
                     aligned16(d.in) && aligned16(d.g);
    const int gran = b16 ? 128 : 16;
    long long sp = (M + nsplit - 1) / nsplit;
    sp = (sp + gran - 1) / gran * gran;
    nsplit = (M + sp - 1) / sp;
    d.nsplit = (int)nsplit;
    d.split_pixels = (int)sp;
    d.ntaps = 9; d.Nout = 16;
    d.partial = ar.floats(slab * nsplit);
    if (!d.partial) return fail("wgrad: workspace too small (%zu floats needed)", slab * nsplit);
    // one 16-channel quad of the coarse side per block (blockIdx.y)
    if (b16) up16_wgrad_bf16_kernel<<<dim3((unsigned)nsplit, KQ, 1), 256, 0, st>>>(d);
    else BY_DT(d.dt, (up16_wgrad_kernel<T, 1><<<dim3((unsigned)nsplit, KQ, 1), 256, 0, st>>>(d)));
    CHECK_LAUNCH("up16_wgrad_kernel");
    ReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    ReduceDesc &r = rb.r[0];
    r.partial = d.partial; r.dw = dw; r.db = db;
    r.nsplit = d.nsplit; r.ntaps = 9; r.K = d.K; r.Nout = 16;
    for (int t = 0; t < 9; ++t) r.woff[t] = t;
    r.wsk = wsk; r.wsn = wsn; r.bias_taps = 1u; r.tail = d.K; r.kside = kside;
    const long long tot = (long long)9 * d.K * 16 + (kside ? d.K : 16);
    reduce_partials_kernel<<<dim3((unsigned)((tot + 31) / 32), 1, 1), 256, 0, st>>>(rb);
    CHECK_LAUNCH("reduce_partials_kernel");
    return 0;
}

unsigned grid_for(long long work_items, int per_block = 256, unsigned cap = 8192)
{
    long long g = (work_items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Blocks of the row-structured element-wise kernels (elt_rows_kernel, bn_bwd_apply_rows_kernel) for
// `steps` block steps of rows: large tensors get 1536 blocks (six per CU) so that every thread makes
// whole trips of ROWS_U rows; small ones one block per step.
unsigned rows_grid(long long steps)
{
    if (steps >= 1536LL * ROWS_U) return 1536u;
    return (unsigned)(steps < 1 ? 1 : (steps < 2048 ? steps : 2048));
}

// Whether the element-wise pass takes the row-structured kernel (which can also leave the output's
// "> 0" bits: fp32 tensors, C a multiple of 32).
bool elt_rows_ok(const void *in, const void *out, const Epilogue &epi, int C)
{
    if (C % 4) return false;
    const int C4 = C / 4;
    return C4 <= 256 && 256 % C4 == 0 && aligned16(in) && aligned16(out) &&
           (!epi.res || (epi.res_C % 4 == 0 && epi.res_coff % 4 == 0 && aligned16(epi.res))) &&
           (!epi.mask || (epi.mask_C % 4 == 0 && epi.mask_coff % 4 == 0 && aligned16(epi.mask))) &&
           (!epi.nscale || aligned16(epi.nscale));
}
bool elt_writes_bits(int dt, const void *in, const void *out, const Epilogue &epi, int C)
{
    return dt == 0 && C % 32 == 0 && elt_rows_ok(in, out, epi, C);
}
int launch_elt(int dt, const void *in, void *out, const Epilogue &epi, long long M, int C, int pix_per_image,
               hipStream_t st)
{
    if (C % 4) return fail("element-wise pass needs C %% 4 == 0 (C = %d)", C);
    EltDesc e = {in, out, epi, M, C, pix_per_image};
    const bool rows = elt_rows_ok(in, out, epi, C);
    if ((epi.bits_out || epi.mask_bits) && !(epi.bits_out && !epi.mask_bits && elt_writes_bits(dt, in, out, epi, C)))
        return fail("element-wise pass: mask bits are written by the row-structured fp32 kernel only");
    if (rows) {
        const long long steps = (M * (C / 4) + 255) / 256;
        BY_DT(dt, (elt_rows_kernel<T><<<rows_grid(steps), 256, 0, st>>>(e)));
    } else {
        BY_DT(dt, (elt_kernel<T><<<grid_for(M * (C / 4)), 256, 0, st>>>(e)));
    }
    CHECK_LAUNCH("elt_kernel");
    return 0;
}

// g_out = gy * (mask > 0)
int launch_mask(int dt, const void *gy, const void *mask, void *out, long long M, int C, hipStream_t st)
{
    Epilogue e = kNoEpi;
    e.mask = mask; e.mask_C = C; e.mask_coff = 0;
    return launch_elt(dt, gy, out, e, M, C, 1, st);
}

// Blocks of the vectorised statistics kernel for M rows of C channels (and the rows of its partial
// buffer): enough to fill the chip, at least a few block steps of rows each.
int bn_stats_blocks(long long M, int C)
{
    const int rp = 256 / (C / 4);
    long long g = (M + rp * 4 - 1) / (rp * 4);
    if (g > 512) g = 512;
    return g < 1 ? 1 : (int)g;
}
size_t bn_scratch_bytes(long long M, int C)
{
    return sizeof(double) * 2 * (size_t)C * ((size_t)bn_stats_blocks(M, C) + 1) + 512;
}

// sums[0..2C) of a statistics pass.  C % 4 == 0 with 16-byte aligned tensors: per-block partial rows
// and a fixed-order reduction; otherwise the scalar kernel with double atomics.
int bn_stats_launch(int dt, const BnStatsDesc &s, bool bwd, Arena &ar, hipStream_t st)
{
    const int C = s.C;
    const bool vec = C % 4 == 0 && aligned16(s.x) && (!bwd || (aligned16(s.gy) && aligned16(s.mean) && aligned16(s.invstd) &&
                     (!s.mask || aligned16(s.mask)) && (!s.nscale || aligned16(s.nscale))));
    if (vec) {
        const int G = bn_stats_blocks(s.M, C);
        double *partial = (double *)ar.take(sizeof(double) * 2 * (size_t)C * G);
        if (!partial) return fail("workspace too small for BatchNorm partial sums");
        BnStats4Desc d4 = {s, partial};
        if (bwd) BY_DT(dt, (bn_stats4_kernel<true, T><<<G, 256, 0, st>>>(d4)));
        else BY_DT(dt, (bn_stats4_kernel<false, T><<<G, 256, 0, st>>>(d4)));
        CHECK_LAUNCH("bn_stats4_kernel");
        bn_reduce_kernel<<<(2 * C + 7) / 8, 256, 0, st>>>(partial, s.sums, G, 2 * C);
        CHECK_LAUNCH("bn_reduce_kernel");
        return 0;
    }
    hipMemsetAsync(s.sums, 0, sizeof(double) * 2 * C, st);
    const unsigned grid = (unsigned)((s.M + s.rows_per_block - 1) / s.rows_per_block);
    if (bwd) BY_DT(dt, (bn_stats_kernel<true, T><<<grid, 256, 0, st>>>(s)));
    else BY_DT(dt, (bn_stats_kernel<false, T><<<grid, 256, 0, st>>>(s)));
    CHECK_LAUNCH("bn_stats_kernel");
    return 0;
}

// Batch statistics of pre [M][C] -> bn->save_mean / save_invstd (+ running stats).
// `conv_partial` (conv_rows > 0): the producing convolution already left that many per-block partial rows.
int bn_train_stats(int dt, const erf_bn_t *bn, const void *pre, long long M, int C, Arena &ar, hipStream_t st,
                   const double *conv_partial = nullptr, int conv_rows = 0)
{
    if (C > 256) return fail("BatchNorm kernels support C <= 256 (C = %d)", C);
    if (!bn->save_mean || !bn->save_invstd) return fail("BatchNorm train mode needs save_mean/save_invstd");
    if (conv_partial && conv_rows > 0) {
        bn_reduce_finalize_kernel<<<(C + 7) / 8, 256, 0, st>>>(conv_partial, conv_rows, bn->save_mean, bn->save_invstd,
                                                              bn->running_mean, bn->running_var,
                                                              bn->num_batches_tracked, M, C, bn->momentum, bn->eps);
        CHECK_LAUNCH("bn_reduce_finalize_kernel");
        return 0;
    }
    if (C % 4 == 0 && aligned16(pre)) {
        // per-block partial rows, then one launch that adds them in a fixed order and finalises
        const int G = bn_stats_blocks(M, C);
        double *partial = (double *)ar.take(sizeof(double) * 2 * (size_t)C * G);
        if (!partial) return fail("workspace too small for BatchNorm partial sums");
        BnStatsDesc s = {pre, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, C, 1, 2048};
        BnStats4Desc d4 = {s, partial};
        BY_DT(dt, (bn_stats4_kernel<false, T><<<G, 256, 0, st>>>(d4)));
        CHECK_LAUNCH("bn_stats4_kernel");
        bn_reduce_finalize_kernel<<<(C + 7) / 8, 256, 0, st>>>(partial, G, bn->save_mean, bn->save_invstd,
                                                              bn->running_mean, bn->running_var,
                                                              bn->num_batches_tracked, M, C, bn->momentum, bn->eps);
        CHECK_LAUNCH("bn_reduce_finalize_kernel");
        return 0;
    }
    double *sums = (double *)ar.take(sizeof(double) * 2 * C);
    if (!sums) return fail("workspace too small for BatchNorm sums");
    const int rows = 2048;
    BnStatsDesc s = {pre, nullptr, nullptr, nullptr, nullptr, nullptr, sums, M, C, 1, rows};
    if (bn_stats_launch(dt, s, false, ar, st)) return 1;
    bn_finalize_kernel<<<(C + 63) / 64, 64, 0, st>>>(sums, bn->save_mean, bn->save_invstd, bn->running_mean,
                                                    bn->running_var, bn->num_batches_tracked, M, C, bn->momentum,
                                                    bn->eps);
    CHECK_LAUNCH("bn_finalize_kernel");
    return 0;
}

// BatchNorm backward on g = gy*(mask>0)*nscale: dx (may alias nothing), dgamma, dbeta.
int bn_backward(int dt, const erf_bn_t *bn, const void *pre, const void *gy, const void *mask,
                const float *nscale, void *dx, float *dgamma, float *dbeta, long long M, int C,
                int pix_per_image, Arena &ar, hipStream_t st)
{
    if (C > 256 || C % 4) return fail("BatchNorm kernels support C <= 256, C %% 4 == 0 (C = %d)", C);
    if (!pre) return fail("BatchNorm backward needs the saved pre-normalisation tensor");
    if (!bn->save_mean || !bn->save_invstd) return fail("BatchNorm backward needs save_mean/save_invstd");
    double *sums = (double *)ar.take(sizeof(double) * 2 * C);
    if (!sums) return fail("workspace too small for BatchNorm sums");
    const int rows = 2048;
    BnStatsDesc s = {pre, gy, mask, nscale, bn->save_mean, bn->save_invstd, sums, M, C, pix_per_image, rows};
    if (bn_stats_launch(dt, s, true, ar, st)) return 1;
    BnBwdDesc b = {pre, gy, mask, nscale, bn->gamma, bn->save_mean, bn->save_invstd, sums,
                   dx, dgamma, dbeta, M, C, pix_per_image, bn->mode == 2};
    const int C4 = C / 4;
    if (256 % C4 == 0 && aligned16(pre) && aligned16(gy) && aligned16(dx) && (!mask || aligned16(mask)) &&
        (!nscale || aligned16(nscale))) {
        const long long steps = (M * C4 + 255) / 256;
        BY_DT(dt, (bn_bwd_apply_rows_kernel<T><<<rows_grid(steps), 256, 0, st>>>(b)));
    } else {
        BY_DT(dt, (bn_bwd_apply_kernel<T><<<grid_for(M * (C / 4)), 256, 0, st>>>(b)));
    }
    CHECK_LAUNCH("bn_bwd_apply_kernel");
    return 0;
}

// In eval mode the backward formulas read mean/invstd from save_*; fill them from running stats.
int bn_eval_prepare(const erf_bn_t *bn, int C, hipStream_t st)
{
    if (!bn->save_mean || !bn->save_invstd) return 0;   // inference without backward
    bn_eval_prepare_kernel<<<(C + 63) / 64, 64, 0, st>>>(bn->running_mean, bn->running_var, bn->save_mean,
                                                        bn->save_invstd, C, bn->eps);
    CHECK_LAUNCH("bn_eval_prepare_kernel");
    return 0;
}

void zero_desc(IgemmDesc &d)
{
    memset(&d, 0, sizeof(d));
    d.ngroups = 1;
    d.os = 1;
    d.ss = 1;
}

// 1-D convolution of non_bottleneck_1d: along H (axis 0) or W (axis 1), 3 taps, dilation dil.
// transpose = data gradient (taps mirrored, weight indices swapped).
void conv1d_desc(IgemmDesc &d, int dt, const void *in, const float *w, void *out, int N, int C, int H, int W,
                 int axis, int dil, bool transpose)
{
    zero_desc(d);
    d.dt = dt;
    d.in = in; d.w = w; d.out = out;
    d.N = N; d.GH = H; d.GW = W;
    d.in_H = H; d.in_W = W; d.in_C = C; d.in_coff = 0; d.K = C;
    d.out_H = H; d.out_W = W; d.out_C = C; d.out_coff = 0; d.Nout = C;
    d.gtap0[0] = 0; d.gntaps[0] = 3;
    for (int t = 0; t < 3; ++t) {
        const int off = (transpose ? -(t - 1) : (t - 1)) * dil;
        d.tdh[t] = axis == 0 ? off : 0;
        d.tdw[t] = axis == 1 ? off : 0;
        d.woff[t] = t;
    }
    if (!transpose) { d.wsk = 3; d.wsn = C * 3; }       // W[co][ci][t]: k = ci, j = co
    else { d.wsk = C * 3; d.wsn = 3; }                  // k = co, j = ci
}

void wgrad1d_desc(WgradDesc &d, int dt, const void *in, const void *g, int N, int C, int H, int W, int axis, int dil)
{
    memset(&d, 0, sizeof(d));
    d.dt = dt;
    d.in = in; d.g = g;
    d.N = N; d.GH = H; d.GW = W;
    d.in_H = H; d.in_W = W; d.in_C = C; d.K = C;
    d.ss = 1; d.os = 1; d.ntaps = 3;
    for (int t = 0; t < 3; ++t) {
        d.tdh[t] = axis == 0 ? (t - 1) * dil : 0;
        d.tdw[t] = axis == 1 ? (t - 1) * dil : 0;
    }
    d.g_H = H; d.g_W = W; d.g_C = C; d.Nout = C;
}

// Tap tables of a 3x3 stride-2 transposed convolution seen from the output-parity classes
// (a,b) = (oh&1, ow&1): kh = 1 for a = 0 (source row i), kh in {0 (source i+1), 2 (source i)} for a = 1.
struct ParityTaps {
    int ntaps, kh[9], kw[9], dh[9], dw[9], oa[9], ob[9];
    int gtap0[4], gntaps[4], goa[4], gob[4];
    unsigned rep_mask;    // one tap per class with zero source offset
};
ParityTaps parity_taps()
{
    ParityTaps p;
    memset(&p, 0, sizeof(p));
    int t = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const int g = a * 2 + b;
            p.gtap0[g] = t; p.goa[g] = a; p.gob[g] = b;
            for (int kh = 0; kh < 3; ++kh) {
                if ((a == 0) != (kh == 1)) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    if ((b == 0) != (kw == 1)) continue;
                    p.kh[t] = kh; p.kw[t] = kw;
                    p.dh[t] = kh == 0 ? 1 : 0; p.dw[t] = kw == 0 ? 1 : 0;
                    p.oa[t] = a; p.ob[t] = b;
                    if (p.dh[t] == 0 && p.dw[t] == 0) p.rep_mask |= 1u << t;
                    ++t;
                }
            }
            p.gntaps[g] = t - p.gtap0[g];
        }
    p.ntaps = t;
    return p;
}

}  // namespace

// Scratch of erf_cross_entropy_forward_static: the block partial sums of the loss live in a device
// array of the library's code object instead of a caller workspace (for callers that may not allocate).
#define ERF_CE_STATIC_BYTES (1u << 20)
static __device__ __align__(256) unsigned char scratch[ERF_CE_STATIC_BYTES];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[kMaxDevices];        // device address of `scratch` on each GPU (0 = not resolved yet)

// Device address of this GPU's copy of `scratch`.  The array's name is only a host-side handle:
// every use of the memory (kernel arguments, memsets, copies) goes through the resolved address.
int static_scratch(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail("static scratch: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail("static scratch: device ordinal %d not supported", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail("static scratch: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
bool static_scratch_resolved()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[dev] != nullptr;
}
}  // namespace

namespace {
// Walks the caller's tensor lists in groups of OPT_MAX_TENSORS; one launch per group.
template <int KIND>
int opt_launch(const char *what, int count, float *const *p, const float *const *g, float *const *m, float *const *v,
               const long long *numel, OptDesc d, hipStream_t st)
{
    if (count < 0) return fail("%s: negative tensor count", what);
    if (count && (!p || !g || !numel || (KIND != OPT_EMA && !m) || (KIND == OPT_ADAM && !v)))
        return fail("%s: null pointer table", what);
    int i = 0;
    while (i < count) {
        d.count = 0;
        d.aligned = 0;
        unsigned long long chunks = 0;
        for (; i < count && d.count < OPT_MAX_TENSORS; ++i) {
            if (numel[i] < 0) return fail("%s: tensor %d has a negative size", what, i);
            if (numel[i] == 0) continue;
            if (!p[i] || !g[i] || (KIND != OPT_EMA && !m[i]) || (KIND == OPT_ADAM && !v[i]))
                return fail("%s: tensor %d has a null pointer", what, i);
            const unsigned long long c = ((unsigned long long)numel[i] + OPT_CHUNK - 1) / OPT_CHUNK;
            if (chunks + c >= (1ull << 31)) {
                if (d.count == 0) return fail("%s: tensor %d is too large", what, i);
                break;                                      // goes into the next launch
            }
            const int k = d.count++;
            d.p[k] = p[i]; d.g[k] = g[i];
            d.m[k] = KIND != OPT_EMA ? m[i] : nullptr;
            d.v[k] = KIND == OPT_ADAM ? v[i] : nullptr;
            d.n[k] = numel[i];
            chunks += c;
            d.end_chunk[k] = (unsigned)chunks;
            const uintptr_t bits = (uintptr_t)d.p[k] | (uintptr_t)d.g[k] | (uintptr_t)d.m[k] | (uintptr_t)d.v[k];
            if ((bits & 15) == 0) d.aligned |= 1u << k;
        }
        if (d.count == 0) continue;
        opt_step_kernel<KIND><<<(unsigned)chunks, OPT_BLOCK, 0, st>>>(d);
        CHECK_LAUNCH("opt_step_kernel");
    }
    return 0;
}
}  // namespace

extern "C" {

int erf_version(void)
{
    return ERF_ABI_VERSION;
}

int erf_set_option(int option, int value)
{
    if (option == ERF_OPT_PAIR_FUSION) return g_sw_pair.set(value);
    if (option == ERF_OPT_WGRAD_LINE) return g_sw_wgrad_line.set(value);
    if (option == ERF_OPT_STREAM16) return g_sw_stream16.set(value);
    if (option == ERF_OPT_PAIR16) return g_sw_pair16.set(value);
    if (option == ERF_OPT_LINE2) return g_line2.exchange(value ? 1 : 0);
    if (option == ERF_OPT_CONV16_ROWS) return g_sw_conv16_rows.set(value);
    if (option == ERF_OPT_X3_ROLL) return g_sw_x3r.set(value);
    if (option == ERF_OPT_POOL_FUSE) return g_sw_pool_fuse.set(value);
    if (option == ERF_OPT_SPLIT_BF16) {
        const int prev = x3_mode();
        g_x3_mode.store(value < 0 ? 0 : (value > 2 ? 2 : value), std::memory_order_relaxed);
        return prev;
    }
    return -1;
}
const char *erf_last_error(void) { return g_err; }
unsigned long long erf_launch_count(void) { return g_launches.load(); }

int erf_batch_begin(void)
{
    if (g_batch.mode != BATCH_OFF) return fail("erf_batch_begin: a batch is already open on this thread");
    g_batch.packs.clear();
    g_batch.reds.clear();
    g_batch.mode = BATCH_COLLECT;
    return 0;
}
int erf_batch_pack(void *stream)
{
    if (g_batch.mode != BATCH_COLLECT) return fail("erf_batch_pack: no batch is collecting on this thread");
    g_batch.mode = BATCH_OFF;       // (launch_pack itself is the ordinary path)
    const int rc = g_batch.packs.empty() ? 0 : launch_pack(g_batch.packs.data(), (int)g_batch.packs.size(), (hipStream_t)stream);
    g_batch.packs.clear();
    if (rc) return 1;
    g_batch.mode = BATCH_RUN;
    return 0;
}
int erf_batch_end(void *stream)
{
    if (g_batch.mode != BATCH_RUN) return fail("erf_batch_end: no batch is running on this thread");
    g_batch.mode = BATCH_OFF;
    hipStream_t st = (hipStream_t)stream;
    std::vector<BatchCtx::Red> reds;
    reds.swap(g_batch.reds);
    // runs of equal geometry (the blocks of a run are alike) go into one launch each, in the order deferred
    for (size_t i = 0; i < reds.size();) {
        const ReduceDesc &a = reds[i].r;
        size_t j = i + 1;
        while (j < reds.size() && j - i < (reds[i].vec ? (size_t)ERF_REDUCE_WIDE : (size_t)ERF_WGRAD_BATCH) &&
               reds[j].vec == reds[i].vec && reds[j].nw == reds[i].nw && reds[j].r.Nout == a.Nout && reds[j].r.K == a.K &&
               reds[j].r.kside == a.kside)
            ++j;
        const int n = (int)(j - i);
        if (reds[i].vec) {
            ReduceBatchWide wb;
            memset(&wb, 0, sizeof(wb));
            for (int k = 0; k < n; ++k) wb.r[k] = reds[i + k].r;
            const int wblocks = (int)((reds[i].nw + 127) / 128), bblocks = ((a.kside ? a.K : a.Nout) + 31) / 32;
            reduce_partials4_wide_kernel<<<dim3((unsigned)(wblocks + bblocks), n, 1), 256, 0, st>>>(wb, wblocks);
            CHECK_LAUNCH("reduce_partials4_wide_kernel");
        } else {
            ReduceBatch rb;
            memset(&rb, 0, sizeof(rb));
            for (int k = 0; k < n; ++k) rb.r[k] = reds[i + k].r;
            const long long tot = reds[i].nw + (a.kside ? a.K : a.Nout);
            reduce_partials_kernel<<<dim3((unsigned)((tot + 31) / 32), n, 1), 256, 0, st>>>(rb);
            CHECK_LAUNCH("reduce_partials_kernel");
        }
        i = j;
    }
    return 0;
}
void erf_batch_abort(void)
{
    g_batch.mode = BATCH_OFF;
    g_batch.packs.clear();
    g_batch.reds.clear();
}

size_t erf_workspace_bytes(long long n_pixels, int max_c)
{
    // six gradient/activation temporaries + weight-gradient slabs + BatchNorm sums + slack
    return (size_t)n_pixels * (size_t)max_c * sizeof(float) * 6 + (size_t)(136u << 20);
}

// ================================================================================ Downsampler
static int downsampler_forward_impl(int dt, const void *x, long long sn, long long sc, long long sh, long long sw,
                                    const float *w, const float *b, const erf_bn_t *bn, void *y, void *pre,
                                    int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if ((H & 1) || (W & 1)) return fail("DownsamplerBlock: H and W must be even (got %dx%d)", H, W);
    const int Cc = Cout - Cin, Ho = H / 2, Wo = W / 2;
    if (Cc <= 0) return fail("DownsamplerBlock: noutput must exceed ninput");
    const long long Mo = (long long)N * Ho * Wo;
    const int mode = bn_mode(bn);
    // BatchNorm in train mode (or in eval mode when backward needs the pre-BN tensor) runs as
    // conv/pool -> pre, [statistics,] normalise + ReLU -> y; otherwise BN folds into the epilogue.
    const bool two_pass = mode == 2 || (mode == 1 && pre);
    void *dst = two_pass ? pre : y;
    if (two_pass && !pre) {
        dst = ar.floats((size_t)Mo * Cout);
        if (!dst) return fail("DownsamplerBlock: workspace too small");
    }
    Epilogue epi = kNoEpi;
    epi.bias = b;
    if (mode == 1 && !two_pass) epi.bn = bn_eval_affine(bn);
    epi.relu = !two_pass;
    if (Cin == 3) {
        if (Cc > 16) return fail("DownsamplerBlock with 3 input channels supports at most 19 outputs");
        // (non-negative strides, an extent a buffer descriptor covers: the kernel's unconditional loads)
        const long long x_ext = (long long)(N - 1) * sn + 2 * sc + (long long)(H - 1) * sh + (long long)(W - 1) * sw + 1;
        const bool x_buf = sn >= 0 && sc >= 0 && sh >= 0 && sw >= 0 && x_ext < (1LL << 29);
        FirstDownDesc d = {x, sn, sc, sh, sw, w, dst, epi, N, H, W, Cc, x_buf ? (unsigned)x_ext : 0u};
        BY_DT(dt, (first_down_fwd_kernel<T><<<(unsigned)((Mo + 255) / 256), 256, 0, st>>>(d)));
        CHECK_LAUNCH("first_down_fwd_kernel");
    } else {
        if (Cin % 4 || Cout % 4) return fail("DownsamplerBlock: channel counts must be multiples of 4");
        if (sc != 1 || sw != Cin || sh != (long long)W * Cin || sn != (long long)H * W * Cin)
            return fail("DownsamplerBlock: input must be NHWC-contiguous");
        IgemmDesc d;
        zero_desc(d);
        d.dt = dt;
        d.in = x; d.w = w; d.out = dst; d.epi = epi;
        d.N = N; d.GH = Ho; d.GW = Wo;
        d.in_H = H; d.in_W = W; d.in_C = Cin; d.K = Cin; d.ss = 2;
        d.out_H = Ho; d.out_W = Wo; d.out_C = Cout; d.Nout = Cc;
        d.gntaps[0] = 9;
        for (int t = 0; t < 9; ++t) { d.tdh[t] = t / 3 - 1; d.tdw[t] = t % 3 - 1; d.woff[t] = t; }
        d.wsk = 9; d.wsn = Cin * 9;
        if (launch_igemm(d, ar, st)) return 1;
        Epilogue pe = epi;
        pe.bias = nullptr;
        PoolDesc p = {x, dst, pe, N, H, W, Cin, Cout, Cc};
        BY_DT(dt, (pool_fwd_kernel<T><<<grid_for(Mo * (Cin / 4), 256, 1u << 30), 256, 0, st>>>(p)));
        CHECK_LAUNCH("pool_fwd_kernel");
    }
    if (two_pass) {
        if (mode == 2) { if (bn_train_stats(dt, bn, dst, Mo, Cout, ar, st)) return 1; }
        else if (bn_eval_prepare(bn, Cout, st)) return 1;
        Epilogue e = kNoEpi;
        e.bn = mode == 2 ? bn_saved_affine(bn) : bn_eval_affine(bn);
        e.relu = 1;
        if (launch_elt(dt, dst, y, e, Mo, Cout, Ho * Wo, st)) return 1;
    }
    return 0;
}

static int downsampler_backward_impl(int dt, const void *x, long long sn, long long sc, long long sh, long long sw,
                                     const float *w, const erf_bn_t *bn, const void *y, const void *pre,
                                     const void *gy, void *dx, float *dw, float *db, float *dgamma,
                                     float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                                     void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    const int Cc = Cout - Cin, Ho = H / 2, Wo = W / 2;
    const long long Mo = (long long)N * Ho * Wo;
    const int mode = bn_mode(bn);
    const bool premasked = (flags & ERF_BWD_GY_PREMASKED) != 0;
    const void *g = gy;
    if (mode || !premasked) {
        float *gbuf = ar.floats((size_t)Mo * Cout);
        if (!gbuf) return fail("DownsamplerBlock backward: workspace too small");
        if (mode) {
            if (bn_backward(dt, bn, pre, gy, premasked ? nullptr : y, nullptr, gbuf, dgamma, dbeta, Mo, Cout,
                            Ho * Wo, ar, st)) return 1;
        } else {
            if (launch_mask(dt, gy, y, gbuf, Mo, Cout, st)) return 1;
        }
        g = gbuf;
    }
    if (Cin == 3) {
        const int nw = Cc * 27;
        // scattered patch loads, twelve in flight per wave: four waves per SIMD hide their latency
        // (48 -> 40 us at batch 6; more blocks than that change nothing: the loads then queue at the
        // texture path); the slabs are 1.5 KB each
        long long nsplit = (Mo + 255) / 256;
        if (nsplit > 1024) nsplit = 1024;
        long long sp = (Mo + nsplit - 1) / nsplit;
        sp = (sp + 15) / 16 * 16;
        nsplit = (Mo + sp - 1) / sp;
        float *partial = ar.floats((size_t)nsplit * (nw + Cc));
        if (!partial) return fail("DownsamplerBlock backward: workspace too small");
        const long long x_elems = (long long)(N - 1) * sn + 2 * sc + (long long)(H - 1) * sh + (long long)(W - 1) * sw + 1;
        if (x_elems >= (1LL << 29) || Mo * Cout >= (1LL << 29) || sn < 0 || sc < 0 || sh < 0 || sw < 0)
            return fail("DownsamplerBlock backward: input too large or negatively strided");
        FirstDownWgradDesc d = {x, sn, sc, sh, sw, g, partial, N, H, W, Cc, Cout, (int)sp, (unsigned)x_elems};
        BY_DT(dt, (first_down_wgrad_kernel<T><<<(unsigned)nsplit, 256, 0, st>>>(d)));
        CHECK_LAUNCH("first_down_wgrad_kernel");
        reduce_flat_kernel<<<(nw + Cc + 31) / 32, 256, 0, st>>>(partial, dw, db, (int)nsplit, nw, Cc);
        CHECK_LAUNCH("reduce_flat_kernel");
    } else {
        WgradDesc d;
        memset(&d, 0, sizeof(d));
        d.dt = dt;
        d.in = x; d.g = g;
        d.N = N; d.GH = Ho; d.GW = Wo;
        d.in_H = H; d.in_W = W; d.in_C = Cin; d.K = Cin; d.ss = 2; d.os = 1; d.ntaps = 9;
        int woff[9];
        for (int t = 0; t < 9; ++t) { d.tdh[t] = t / 3 - 1; d.tdw[t] = t % 3 - 1; woff[t] = t; }
        d.g_H = Ho; d.g_W = Wo; d.g_C = Cout; d.Nout = Cc;
        const bool x_nhwc = sc == 1 && sw == Cin && sh == (long long)W * Cin && sn == (long long)H * W * Cin;
        if (Cin == 16 && x_nhwc && sampler16_shape_ok(Cc, Mo, Cout, Cin)) {
            // coarse side = the gradient of the convolution output, fine side = the block input
            WgradDesc e;
            memset(&e, 0, sizeof(e));
            e.dt = dt;
            e.in = g; e.in_C = Cout; e.K = Cc;
            e.g = x; e.g_C = Cin;
            e.N = N; e.GH = Ho; e.GW = Wo;
            if (launch_up16_wgrad(e, dw, db, Cin * 9, 9, 1, ar, st)) return 1;
        } else if (launch_wgrad(d, dw, db, woff, 9, Cin * 9, 1u, ar, st)) return 1;
    }
    if (dx) {
        // dx is NHWC [N,H,W,Cin]: max-pool gradient first, then the conv data gradient on top.
        if (Cin == 3 && (sc != 1 || sw != Cin || sh != (long long)W * Cin || sn != (long long)H * W * Cin))
            return fail("DownsamplerBlock backward: dx for a 3-channel input needs an NHWC input");
        const ParityTaps pt = parity_taps();
        IgemmDesc d;
        zero_desc(d);
        d.dt = dt;
        d.in = g; d.w = w; d.out = dx;
        if (flags & ERF_BWD_MASK_DX) { d.epi.mask = x; d.epi.mask_C = Cin; }
        d.N = N; d.GH = Ho; d.GW = Wo;
        d.in_H = Ho; d.in_W = Wo; d.in_C = Cout; d.in_coff = 0; d.K = Cc; d.ss = 1;
        d.out_H = H; d.out_W = W; d.out_C = Cin; d.Nout = Cin; d.os = 2;
        d.ngroups = 4;
        for (int gq = 0; gq < 4; ++gq) {
            d.gtap0[gq] = pt.gtap0[gq]; d.gntaps[gq] = pt.gntaps[gq]; d.goa[gq] = pt.goa[gq]; d.gob[gq] = pt.gob[gq];
        }
        for (int t = 0; t < 9; ++t) { d.tdh[t] = pt.dh[t]; d.tdw[t] = pt.dw[t]; d.woff[t] = pt.kh[t] * 3 + pt.kw[t]; }
        d.wsk = Cin * 9; d.wsn = 9;   // W[co][ci][kh][kw]: k = co, j = ci
        // 16-channel input on the sampler kernel: the pool's gradient is formed in its epilogue (the
        // lane that holds four channels of a window's four pixels reads the window's gradient);
        // otherwise pool_bwd_kernel writes it into dx and the convolution adds to it
        const bool no_pool_fuse = !g_sw_pool_fuse.on();
        const bool x_nhwc16 = sc == 1 && sw == Cin && sh == (long long)W * Cin && sn == (long long)H * W * Cin;
        if (!no_pool_fuse && Cin == 16 && x_nhwc16 && aligned16(x) && aligned16(g) && Cc % 4 == 0 && Cout % 4 == 0 &&
            Cc == 48 && up16_fwd_ok(d)) {       // (the pooled kernel variant is built for 48 convolution channels)
            d.pool_x = x; d.pool_g = g; d.pool_gC = Cout; d.pool_gcoff = Cc;
            return launch_up16_fwd(d, st);
        }
        PoolBwdDesc p;
        p.in = x; p.g = g; p.dx = dx; p.N = N; p.H = H; p.W = W; p.C = Cin; p.g_C = Cout; p.g_coff = Cc;
        BY_DT(dt, (pool_bwd_kernel<T><<<grid_for(Mo * Cin, 256, 1u << 30), 256, 0, st>>>(p)));
        CHECK_LAUNCH("pool_bwd_kernel");
        d.epi.res = dx; d.epi.res_C = Cin; d.epi.res_coff = 0;
        if (up16_fwd_ok(d)) {
            if (launch_up16_fwd(d, st)) return 1;
        } else if (launch_igemm(d, ar, st)) return 1;
    }
    return 0;
}

// ================================================================================ non_bottleneck_1d
// Forward of a chain of non_bottleneck_1d blocks (y_0 = block_0(x), y_i = block_i(y_{i-1})).
// Launch plan per block: conv3x1_1 | conv1x3_1 + conv3x1_2 (pair over image columns) | conv1x3_2;
// inside a chain a block's conv1x3_2 and the next block's conv3x1_1 form one more such pair
// ("along W, then along H"), whose intermediate is the block output.
// If `d` (planned) runs on line_x3_kernel and has no mask / residual / ReLU epilogue, reserve its
// statistics rows and return their number; else 0 (ERFNET_NO_BN_FUSE=1: never — experiments).
int conv_stats_rows(IgemmDesc &d, Arena &ar)
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("ERFNET_NO_BN_FUSE"); off = (e && atoi(e)) ? 1 : 0; }
    d.bn_partial = nullptr;
    // (128 channels only: at 64 channels, two blocks per CU, the extra state spills and the fused kernel
    // costs more than the statistics pass it saves — measured 94 us against 45 + 9)
    if (off || d.cfg != kCfgLineX3 || d.K != 128 || d.epi.mask || d.epi.res || d.epi.relu || d.epi.bn.on) return 0;
    const int rows = line_x3_grid(d);
    if (rows <= 0) return 0;
    d.bn_partial = (double *)ar.take(sizeof(double) * 2 * (size_t)d.Nout * rows);
    return d.bn_partial ? rows : 0;
}

namespace {
struct Nb1dPlan {
    IgemmDesc d[4];
    int m1, m2;
    bool tp1, tp2;
    void *p2, *t2, *p4;
    int pair12, pair_next;      // tile variants, 0 = separate launches
    FusedPair pd12, pdn;
};
constexpr int kMaxChain = 8;

int nb1d_chain_segment(int dt, const void *x, const erf_nb1d_block_t *blk, int nb, int N, int C, int H, int W,
                       Arena &ar, hipStream_t st)
{
    const long long M = (long long)N * H * W;
    const size_t nel = (size_t)M * C;
    static thread_local Nb1dPlan plan[kMaxChain];
    PackDesc pk[kMaxChain * 4];
    float *wa = nullptr, *wc = nullptr;          // scratch tensors of blocks that keep nothing
    for (int i = 0; i < nb; ++i) {
        const erf_nb1d_block_t &B = blk[i];
        Nb1dPlan &P = plan[i];
        if (B.dil < 1) return fail("non_bottleneck_1d: dilation must be >= 1");
        const void *xin = i ? blk[i - 1].y : x;
        P.m1 = bn_mode(B.bn1); P.m2 = bn_mode(B.bn2);
        // Two-pass BatchNorm (conv -> pre, [statistics,] normalise) in train mode and whenever the
        // activations are saved for backward; eval-mode inference folds BN into the conv epilogue.
        P.tp1 = P.m1 == 2 || (P.m1 == 1 && B.save);
        P.tp2 = P.m2 == 2 || (P.m2 == 1 && B.save);
        void *t1, *t3;
        if (B.save) {
            t1 = B.save[0]; P.p2 = B.save[1]; P.t2 = B.save[2]; t3 = B.save[3]; P.p4 = B.save[4];
            if ((P.m1 && !P.p2) || (P.m2 && !P.p4)) return fail("non_bottleneck_1d: BatchNorm needs the pre2/pre4 buffers");
        } else {
            if (!wa) { wa = ar.floats(nel); wc = ar.floats(nel); }
            if (!wa || !wc) return fail("non_bottleneck_1d: workspace too small");
            t1 = wa; P.p2 = wc; P.t2 = wc; t3 = wa; P.p4 = wc;
        }
        IgemmDesc *d = P.d;
        // conv3x1_1 + ReLU
        conv1d_desc(d[0], dt, xin, B.w[0], t1, N, C, H, W, 0, 1, false);
        d[0].epi.bias = B.b[0]; d[0].epi.relu = 1;
        // conv1x3_1 [+ BN1] + ReLU
        conv1d_desc(d[1], dt, t1, B.w[1], P.tp1 ? P.p2 : P.t2, N, C, H, W, 1, 1, false);
        d[1].epi.bias = B.b[1];
        if (!P.tp1) {
            if (P.m1 == 1) d[1].epi.bn = bn_eval_affine(B.bn1);
            d[1].epi.relu = 1;
        }
        // conv3x1_2 (dilated) + ReLU
        conv1d_desc(d[2], dt, P.t2, B.w[2], t3, N, C, H, W, 0, B.dil, false);
        d[2].epi.bias = B.b[2]; d[2].epi.relu = 1;
        // conv1x3_2 (dilated) [+ BN2] [+ dropout] + residual + ReLU
        conv1d_desc(d[3], dt, t3, B.w[3], P.tp2 ? P.p4 : B.y, N, C, H, W, 1, B.dil, false);
        d[3].epi.bias = B.b[3];
        if (!P.tp2) {
            if (P.m2 == 1) d[3].epi.bn = bn_eval_affine(B.bn2);
            d[3].epi.nscale = B.drop_scale;
            d[3].epi.res = xin; d[3].epi.res_C = C; d[3].epi.res_coff = 0;
            d[3].epi.relu = 1;
        }
        P.pair12 = P.tp1 ? 0 : pair_variant(d[1], d[2]);
        if (P.pair12 && !B.save) {
            // the pair reads neighbouring columns of t1 while other blocks write their columns of
            // t3: the two must not share a buffer; the intermediate's buffer is free (it stays in LDS)
            d[2].out = P.p2;
            d[3].in = P.p2;
        }
        P.pair_next = 0;
    }
    // conv1x3_2 of block i with conv3x1_1 of block i+1 (their buffers must not alias)
    for (int i = 0; i + 1 < nb; ++i) {
        Nb1dPlan &P = plan[i];
        const IgemmDesc &a = P.d[3], &b = plan[i + 1].d[0];
        if (P.tp2 || b.out == a.in) continue;
        P.pair_next = pair_variant(a, b);
    }
    for (int i = 0; i < nb; ++i) {
        Nb1dPlan &P = plan[i];
        PackDesc *k = pk + 4 * i;
        // (a block whose conv3x1_1 runs in the previous block's pair got its slot 0 filled there)
        if (!(i > 0 && plan[i - 1].pair_next) && plan_igemm(P.d[0], ar, k[0])) return 1;
        if (P.pair12) {
            if (plan_pair(P.d[1], P.d[2], blk[i].save != nullptr, ar, k + 1, P.pd12)) return 1;
        } else if (plan_igemm(P.d[1], ar, k[1]) || plan_igemm(P.d[2], ar, k[2])) return 1;
        if (P.pair_next) {
            PackDesc two[2];
            if (plan_pair(P.d[3], plan[i + 1].d[0], true, ar, two, P.pdn)) return 1;
            k[3] = two[0];
            k[4] = two[1];          // slot 0 of the next block
        } else if (plan_igemm(P.d[3], ar, k[3])) return 1;
    }
    if (launch_pack(pk, 4 * nb, st)) return 1;             // one launch repacks the weights of the whole run
    // ReLU masks as bits: out of the epilogues of the launches that end in the ReLU and run on the
    // rolling kernel (not the paired launches, not a convolution followed by a BatchNorm pass)
    for (int i = 0; i < nb; ++i) {
        const erf_nb1d_block_t &B = blk[i];
        Nb1dPlan &P = plan[i];
        unsigned written = 0;
        if (B.sign_bits) {
            const bool own[4] = {!(i > 0 && plan[i - 1].pair_next), !P.pair12 && !P.tp1, !P.pair12,
                                 !P.pair_next && !P.tp2};
            for (int k = 0; k < 4; ++k) {
                if (!B.sign_bits[k] || !own[k] || !P.d[k].epi.relu || !igemm_on_x3r(P.d[k])) continue;
                P.d[k].epi.bits_out = B.sign_bits[k];
                written |= 1u << k;
            }
        }
        if (B.sign_bits_written) *B.sign_bits_written = written;
    }
    // bf16 tensors: the eight-wave pair kernel writes the bits of both of its outputs (conv1x3_1 +
    // conv3x1_2 -> t2, t3; conv1x3_2 + the next block's conv3x1_1 -> y, the next block's t1)
    for (int i = 0; dt && i < nb; ++i) {
        const erf_nb1d_block_t &B = blk[i];
        Nb1dPlan &P = plan[i];
        auto give = [&](Epilogue &e, const erf_nb1d_block_t &owner, int k) {
            if (!owner.sign_bits || !owner.sign_bits[k] || !owner.sign_bits_written || !e.relu) return;
            e.bits_out = owner.sign_bits[k];
            *owner.sign_bits_written |= 1u << k;
        };
        if (P.pair12 && !P.pd12.p16_on && B.save && line2_takes(P.pd12.ld, C, P.pair12, true)) {
            give(P.pd12.ld.epi1, B, 1);
            give(P.pd12.ld.epi2, B, 2);
        }
        if (P.pair_next && !P.pdn.p16_on && line2_takes(P.pdn.ld, C, P.pair_next, true)) {
            give(P.pdn.ld.epi1, B, 3);
            give(P.pdn.ld.epi2, blk[i + 1], 0);
        }
    }

    for (int i = 0; i < nb; ++i) {
        const erf_nb1d_block_t &B = blk[i];
        Nb1dPlan &P = plan[i];
        const void *xin = i ? blk[i - 1].y : x;
        if (!(i > 0 && plan[i - 1].pair_next) && run_igemm(P.d[0], st)) return 1;
        if (P.pair12) {
            if (run_pair(P.pd12, C, P.pair12, st)) return 1;
        } else {
            // train-mode BatchNorm behind a convolution on line_x3_kernel: the kernel leaves the column
            // sums of its output, the separate statistics pass is skipped
            const int g1 = P.m1 == 2 ? conv_stats_rows(P.d[1], ar) : 0;
            if (run_igemm(P.d[1], st)) return 1;
            if (P.tp1) {
                if (P.m1 == 2) { if (bn_train_stats(dt, B.bn1, P.p2, M, C, ar, st, P.d[1].bn_partial, g1)) return 1; }
                else if (bn_eval_prepare(B.bn1, C, st)) return 1;
                Epilogue e = kNoEpi;
                e.bn = P.m1 == 2 ? bn_saved_affine(B.bn1) : bn_eval_affine(B.bn1);
                e.relu = 1;
                if (B.sign_bits && B.sign_bits[1] && B.sign_bits_written && elt_writes_bits(dt, P.p2, P.t2, e, C)) {
                    e.bits_out = B.sign_bits[1];        // t2 = relu(bn1(.)): the normalisation pass writes its mask
                    *B.sign_bits_written |= 2u;
                }
                if (launch_elt(dt, P.p2, P.t2, e, M, C, H * W, st)) return 1;
            }
            if (run_igemm(P.d[2], st)) return 1;
        }
        if (P.pair_next) {
            if (run_pair(P.pdn, C, P.pair_next, st)) return 1;
            continue;
        }
        const int g2 = (P.tp2 && P.m2 == 2) ? conv_stats_rows(P.d[3], ar) : 0;
        if (run_igemm(P.d[3], st)) return 1;
        if (P.tp2) {
            if (P.m2 == 2) { if (bn_train_stats(dt, B.bn2, P.p4, M, C, ar, st, P.d[3].bn_partial, g2)) return 1; }
            else if (bn_eval_prepare(B.bn2, C, st)) return 1;
            Epilogue e = kNoEpi;
            e.bn = P.m2 == 2 ? bn_saved_affine(B.bn2) : bn_eval_affine(B.bn2);
            e.nscale = B.drop_scale;
            e.res = xin; e.res_C = C; e.res_coff = 0;
            e.relu = 1;
            if (B.sign_bits && B.sign_bits[3] && B.sign_bits_written && elt_writes_bits(dt, P.p4, B.y, e, C)) {
                e.bits_out = B.sign_bits[3];
                *B.sign_bits_written |= 8u;
            }
            if (launch_elt(dt, P.p4, B.y, e, M, C, H * W, st)) return 1;
        }
    }
    return 0;
}
}  // namespace

static int nb1d_chain_forward_impl(int dt, const void *x, const erf_nb1d_block_t *blocks, int nblocks, int N, int C,
                                   int H, int W, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (C % 4) return fail("non_bottleneck_1d: chann must be a multiple of 4 (got %d)", C);
    if (nblocks < 1 || !blocks) return fail("non_bottleneck_1d chain: no blocks");
    for (int i = 0; i < nblocks; ++i)
        if (!blocks[i].y) return fail("non_bottleneck_1d chain: block %d has no output buffer", i);
    // segments of kMaxChain blocks; the stream orders a segment's kernels behind the previous
    // segment's, so each segment may reuse the workspace from its start
    for (int i = 0; i < nblocks; i += kMaxChain) {
        Arena ar(ws, ws_bytes);
        const int nb = nblocks - i < kMaxChain ? nblocks - i : kMaxChain;
        if (nb1d_chain_segment(dt, i ? blocks[i - 1].y : x, blocks + i, nb, N, C, H, W, ar, st)) return 1;
    }
    return 0;
}

static int nb1d_forward_impl(int dt, const void *x, const float *const w[4], const float *const b[4],
                             const erf_bn_t *bn1, const erf_bn_t *bn2, const float *drop_scale, int dil, void *y,
                             void *const save[5], int N, int C, int H, int W, void *ws, size_t ws_bytes,
                             void *stream)
{
    erf_nb1d_block_t blk;
    for (int i = 0; i < 4; ++i) { blk.w[i] = w[i]; blk.b[i] = b[i]; }
    blk.bn1 = bn1; blk.bn2 = bn2; blk.drop_scale = drop_scale; blk.dil = dil; blk.y = y; blk.save = save;
    blk.sign_bits = nullptr; blk.sign_bits_written = nullptr;
    return nb1d_chain_forward_impl(dt, x, &blk, 1, N, C, H, W, ws, ws_bytes, stream);
}

// ================================================================================ convolution pair
int erf_conv_pair_pack(const float *w_first, const float *w_second, int C, float *packed, void *stream)
{
    if (C != 64 && C != 128) return fail("conv pair: 64 or 128 channels (got %d)", C);
    if (!w_first || !w_second || !packed) return fail("conv pair: null weight pointer");
    PackDesc pk[2];
    const float *w[2] = {w_first, w_second};
    for (int i = 0; i < 2; ++i) {
        memset(&pk[i], 0, sizeof(PackDesc));
        pk[i].w = w[i]; pk[i].wp = packed + (size_t)i * 3 * C * C;
        pk[i].ntaps = 3; pk[i].K = C; pk[i].Nout = C; pk[i].Npad = C; pk[i].wsk = 3; pk[i].wsn = C * 3; pk[i].kc = 16;
        for (int t = 0; t < 3; ++t) pk[i].woff[t] = t;
    }
    return launch_pack(pk, 2, (hipStream_t)stream);
}

int erf_conv_pair_forward(const float *x, const float *packed, const float *b_first, const float *b_second,
                          int first_axis, int dil_first, int dil_second, int relu_first, int relu_second,
                          const float *residual, float *mid, float *y, int N, int C, int H, int W,
                          void *stream)
{
    if (first_axis != 0 && first_axis != 1) return fail("conv pair: first_axis is 0 (H) or 1 (W)");
    if (dil_first < 1 || dil_second < 1) return fail("conv pair: dilation must be >= 1");
    if (!x || !packed || !y) return fail("conv pair: null tensor pointer");
    if (y == x || mid == x) return fail("conv pair: the outputs must not alias the input");
    IgemmDesc a, b;
    // descriptors only for the geometry checks of pair_variant; an unstored intermediate is
    // given the input's (aligned) address there
    float *link = mid ? mid : const_cast<float *>(x);
    conv1d_desc(a, 0, x, nullptr, link, N, C, H, W, first_axis, dil_first, false);
    conv1d_desc(b, 0, link, nullptr, y, N, C, H, W, 1 - first_axis, dil_second, false);
    a.epi.bias = b_first; a.epi.relu = relu_first ? 1 : 0;
    b.epi.bias = b_second; b.epi.relu = relu_second ? 1 : 0;
    if (residual) { b.epi.res = residual; b.epi.res_C = C; }
    const int tile = pair_variant(a, b, true);
    if (!tile) return fail("conv pair: %d channels with lines of %d pixels have no fused kernel", C, first_axis ? H : W);
    PairDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.in = x; pd.wp1 = packed; pd.wp2 = packed + (size_t)3 * C * C;
    pd.mid = mid; pd.out = y;
    pd.epi1 = a.epi; pd.epi2 = b.epi;
    pd.N = N; pd.img_pixels = H * W;
    if (first_axis == 1) {
        pd.lines_per_img = W; pd.L = H; pd.pstride = W; pd.qstride = 1;
        for (int t = 0; t < 3; ++t) { pd.s1[t] = a.tdw[t]; pd.s2[t] = b.tdh[t]; }
    } else {
        pd.lines_per_img = H; pd.L = W; pd.pstride = 1; pd.qstride = W;
        for (int t = 0; t < 3; ++t) { pd.s1[t] = a.tdh[t]; pd.s2[t] = b.tdw[t]; }
    }
    pd.nlines = N * pd.lines_per_img;
    return run_pair_f32(pd, C, tile, (hipStream_t)stream);
}

// One 3-tap 1-D convolution C -> C on fp32 tensors through line_x3_kernel (split-bf16 arithmetic),
// whatever ERF_OPT_SPLIT_BF16 says: packed = 3 planes x 3*C*C bf16.
int erf_conv_line_pack_split(const float *w, int C, void *packed, void *stream)
{
    if (C != 64 && C != 128) return fail("conv line (split): 64 or 128 channels (got %d)", C);
    if (!w || !packed) return fail("conv line (split): null weight pointer");
    LinePackDesc pk;
    const int woff[3] = {0, 1, 2};
    line_pack_desc(pk, w, (bf16_t *)packed, C, 3, C * 3, woff, true);
    return launch_line_pack(&pk, 1, (hipStream_t)stream);
}

int erf_conv_line_forward_split(const float *x, const void *packed, const float *bias, int axis, int dil, int relu,
                                const float *residual, const float *mask, float *y, int N, int C, int H, int W,
                                void *stream)
{
    if (C != 64 && C != 128) return fail("conv line (split): 64 or 128 channels (got %d)", C);
    if (axis != 0 && axis != 1) return fail("conv line (split): axis is 0 (H) or 1 (W)");
    if (dil < 1) return fail("conv line (split): dilation must be >= 1");
    if (!x || !packed || !y) return fail("conv line (split): null tensor pointer");
    if (y == x) return fail("conv line (split): the output must not alias the input");
    if (!aligned16(x) || !aligned16(y) || !aligned16(residual) || !aligned16(mask) || !aligned16(packed))
        return fail("conv line (split): tensors must be 16-byte aligned");
    if ((long long)N * H * W * C >= (1LL << 29)) return fail("conv line (split): tensor too large");
    LineDesc ld;
    memset(&ld, 0, sizeof(ld));
    ld.in = (const bf16_t *)x; ld.wp1 = (const bf16_t *)packed; ld.out = (bf16_t *)y;
    ld.epi1 = kNoEpi; ld.epi2 = kNoEpi;
    ld.epi1.bias = bias; ld.epi1.relu = relu ? 1 : 0;
    if (residual) { ld.epi1.res = residual; ld.epi1.res_C = C; }
    if (mask) { ld.epi1.mask = mask; ld.epi1.mask_C = C; }
    ld.N = N; ld.img_pixels = H * W;
    // the taps run across the lines: a convolution along W takes image columns as lines
    if (axis == 1) { ld.lines_per_img = W; ld.L = H; ld.pstride = W; ld.qstride = 1; }
    else { ld.lines_per_img = H; ld.L = W; ld.pstride = 1; ld.qstride = W; }
    for (int t = 0; t < 3; ++t) ld.s1[t] = (t - 1) * dil;
    ld.nlines = N * ld.lines_per_img;
    return run_line_x3(ld, C, (hipStream_t)stream);
}

int erf_conv_pair_pack_bf16(const float *w_first, const float *w_second, int C, void *packed, void *stream)
{
    if (C != 64 && C != 128) return fail("conv pair (bf16): 64 or 128 channels (got %d)", C);
    if (!w_first || !packed) return fail("conv pair (bf16): null weight pointer");
    LinePackDesc pk[2];
    const int woff[3] = {0, 1, 2};
    line_pack_desc(pk[0], w_first, (bf16_t *)packed, C, 3, C * 3, woff);
    if (w_second) line_pack_desc(pk[1], w_second, (bf16_t *)packed + line_packed_elems(C), C, 3, C * 3, woff);
    return launch_line_pack(pk, w_second ? 2 : 1, (hipStream_t)stream);
}

int erf_conv_pair_forward_bf16(const void *x, const void *packed, const float *b_first, const float *b_second,
                               int first_axis, int dil_first, int dil_second, int relu_first, int relu_second,
                               const void *residual, void *mid, void *y, int N, int C, int H, int W,
                               void *stream)
{
    if (first_axis != 0 && first_axis != 1) return fail("conv pair (bf16): first_axis is 0 (H) or 1 (W)");
    if (dil_first < 1 || dil_second < 0) return fail("conv pair (bf16): bad dilation");
    if (!x || !packed || !y) return fail("conv pair (bf16): null tensor pointer");
    if (y == x || mid == x) return fail("conv pair (bf16): the outputs must not alias the input");
    if (!aligned16(x) || !aligned16(y) || !aligned16(mid) || !aligned16(residual) || !aligned16(packed))
        return fail("conv pair (bf16): tensors must be 16-byte aligned");
    if ((long long)N * H * W * C >= (1LL << 31)) return fail("conv pair (bf16): tensor too large");
    const bool pair = dil_second > 0;
    // the first convolution runs across the lines: along W -> lines are image columns
    const int L = first_axis == 1 ? H : W;
    const int BM = line_tile_bf16(C, L, pair);
    if (!BM) return fail("conv pair (bf16): %d channels with lines of %d pixels have no kernel", C, L);
    LineDesc ld;
    memset(&ld, 0, sizeof(ld));
    ld.in = (const bf16_t *)x; ld.wp1 = (const bf16_t *)packed; ld.wp2 = (const bf16_t *)packed + line_packed_elems(C);
    ld.mid = (bf16_t *)mid; ld.out = (bf16_t *)y;
    ld.epi1 = kNoEpi; ld.epi2 = kNoEpi;
    ld.epi1.bias = b_first; ld.epi1.relu = relu_first ? 1 : 0;
    ld.epi2.bias = b_second; ld.epi2.relu = relu_second ? 1 : 0;
    Epilogue &last = pair ? ld.epi2 : ld.epi1;
    if (residual) { last.res = residual; last.res_C = C; }
    ld.N = N; ld.img_pixels = H * W; ld.L = L;
    ld.segs = (L + BM - 1) / BM;
    if (first_axis == 1) { ld.lines_per_img = W; ld.pstride = W; ld.qstride = 1; }
    else { ld.lines_per_img = H; ld.pstride = 1; ld.qstride = W; }
    for (int t = 0; t < 3; ++t) { ld.s1[t] = (t - 1) * dil_first; ld.s2[t] = (t - 1) * dil_second; }
    ld.nlines = N * ld.lines_per_img;
    return run_line(ld, C, BM, pair, (hipStream_t)stream);
}

static int nb1d_backward_impl(int dt, const void *x, const float *const w[4], const erf_bn_t *bn1, const erf_bn_t *bn2,
                              const float *drop_scale, int dil, const void *y, const void *const save[5],
                              const unsigned *const bits[4],
                              const void *gy, void *dx, float *const dw[4], float *const db[4],
                              float *const dgamma[2], float *const dbeta[2], int N, int C, int H, int W, int flags,
                              void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (C % 4) return fail("non_bottleneck_1d: chann must be a multiple of 4 (got %d)", C);
    if (!save) return fail("non_bottleneck_1d backward needs the saved activations");
    const long long M = (long long)N * H * W;
    const size_t nel = (size_t)M * C;
    const int m1 = bn_mode(bn1), m2 = bn_mode(bn2);
    const void *t1 = save[0], *p2 = save[1], *t2 = save[2], *t3 = save[3], *p4 = save[4];
    // Every tensor of the gradient chain gets a buffer of its own, so the weight gradients (helper
    // stream) can read them while the chain moves on:
    //   g4 = gy*(y>0) -> [gs = bn2'/dropout'(g4)] -> gb -> ga -> [gn = bn1'(ga)] -> g1 -> dx
    const bool premasked = (flags & ERF_BWD_GY_PREMASKED) != 0;
    float *g4buf = premasked ? nullptr : ar.floats(nel);
    const void *g4 = premasked ? gy : g4buf;
    float *ga = ar.floats(nel), *gb = ar.floats(nel), *g1 = ar.floats(nel);
    float *gsbuf = (m2 || drop_scale) ? ar.floats(nel) : nullptr;
    const void *gs = gsbuf ? gsbuf : g4;
    float *gnbuf = m1 ? ar.floats(nel) : nullptr;
    const void *gn = gnbuf ? gnbuf : ga;
    if ((!premasked && !g4buf) || !ga || !gb || !g1 || ((m2 || drop_scale) && !gsbuf) || (m1 && !gnbuf))
        return fail("non_bottleneck_1d backward: workspace too small");
    const int woff[3] = {0, 1, 2};

    IgemmDesc d[4];
    PackDesc pk[4];
    conv1d_desc(d[3], dt, gs, w[3], gb, N, C, H, W, 1, dil, true);      // masked by t3 > 0
    d[3].epi.mask = t3; d[3].epi.mask_C = C;
    conv1d_desc(d[2], dt, gb, w[2], ga, N, C, H, W, 0, dil, true);      // masked by t2 > 0
    d[2].epi.mask = t2; d[2].epi.mask_C = C;
    conv1d_desc(d[1], dt, gn, w[1], g1, N, C, H, W, 1, 1, true);        // masked by t1 > 0
    d[1].epi.mask = t1; d[1].epi.mask_C = C;
    conv1d_desc(d[0], dt, g1, w[0], dx, N, C, H, W, 0, 1, true);        // + g4 (residual path)
    d[0].epi.res = g4; d[0].epi.res_C = C;
    if (flags & ERF_BWD_MASK_DX) { d[0].epi.mask = x; d[0].epi.mask_C = C; }
    // each half of the chain is "along W, then along H": one launch over image columns
    const int pair32 = pair_variant(d[3], d[2]), pair10 = pair_variant(d[1], d[0]);
    FusedPair pd32, pd10;
    if (pair32) {
        PackDesc two[2];
        if (plan_pair(d[3], d[2], true, ar, two, pd32)) return 1;
        pk[3] = two[0]; pk[2] = two[1];
    } else if (plan_igemm(d[3], ar, pk[3]) || plan_igemm(d[2], ar, pk[2])) return 1;
    if (pair10) {
        PackDesc two[2];
        if (plan_pair(d[1], d[0], true, ar, two, pd10)) return 1;
        pk[1] = two[0]; pk[0] = two[1];
    } else if (plan_igemm(d[1], ar, pk[1]) || plan_igemm(d[0], ar, pk[0])) return 1;
    if (g_batch.mode == BATCH_COLLECT) {    // first pass over a run: nothing has been launched yet
        g_batch.packs.insert(g_batch.packs.end(), pk, pk + 4);
        return 0;
    }
    if (g_batch.mode != BATCH_RUN && launch_pack(pk, 4, st)) return 1;
    if (bits) {     // masks given as bits serve the launches of the rolling kernel (t1, t2, t3, x -> d[1], d[2], d[3], d[0])
        const int slot[4] = {3, 0, 1, 2};
        const bool own[4] = {!pair10, !pair10, !pair32, !pair32};
        for (int k = 0; k < 4; ++k) {
            if (!bits[slot[k]] || !d[k].epi.mask || !own[k] || !igemm_on_x3r(d[k])) continue;
            d[k].epi.mask_bits = bits[slot[k]];
            d[k].epi.mask = nullptr; d[k].epi.mask_C = 0;
        }
    }
    if (bits && dt) {   // bf16: the eight-wave pair kernel reads either form, each of its two masks on its own
        auto take = [&](Epilogue &e, int k) {
            if (!bits[k] || !e.mask) return;
            e.mask_bits = bits[k];
            e.mask = nullptr; e.mask_C = 0;
        };
        if (pair32 && !pd32.p16_on && line2_takes(pd32.ld, C, pair32, true)) { take(pd32.ld.epi1, 2); take(pd32.ld.epi2, 1); }
        if (pair10 && !pd10.p16_on && line2_takes(pd10.ld, C, pair10, true)) { take(pd10.ld.epi1, 0); take(pd10.ld.epi2, 3); }
    }
    // BatchNorm scratch first; the slabs of the four weight gradients follow it
    const size_t bn_bytes = 2 * bn_scratch_bytes(M, C);
    void *pbn = ar.take(bn_bytes);
    if (!pbn) return fail("non_bottleneck_1d backward: workspace too small");
    Arena bn_ar(pbn, bn_bytes);
    // g4 = gy * (y > 0): gradient of the residual sum (goes to x unchanged and into the conv branch)
    if (!premasked && launch_mask(dt, gy, y, g4buf, M, C, st)) return 1;
    if (m2) {
        if (bn_backward(dt, bn2, p4, g4, nullptr, drop_scale, gsbuf, dgamma ? dgamma[1] : nullptr,
                        dbeta ? dbeta[1] : nullptr, M, C, H * W, bn_ar, st)) return 1;
    } else if (drop_scale) {
        Epilogue e = kNoEpi;
        e.nscale = drop_scale;
        if (launch_elt(dt, g4, gsbuf, e, M, C, H * W, st)) return 1;
    }
    // data-gradient chain: gs -> gb -> ga [-> gn] -> g1 -> dx = dgrad + g4
    if (pair32) {
        if (run_pair(pd32, C, pair32, st)) return 1;
    } else if (run_igemm(d[3], st) || run_igemm(d[2], st)) return 1;
    if (m1) {
        if (bn_backward(dt, bn1, p2, ga, nullptr, nullptr, gnbuf, dgamma ? dgamma[0] : nullptr,
                        dbeta ? dbeta[0] : nullptr, M, C, H * W, bn_ar, st)) return 1;
    }
    if (pair10) {
        if (run_pair(pd10, C, pair10, st)) return 1;
    } else if (run_igemm(d[1], st) || run_igemm(d[0], st)) return 1;
    // the four weight gradients in one launch: (x, g1), (t1, gn), (t2, gb), (t3, gs)
    WgradDesc wd[4];
    wgrad1d_desc(wd[0], dt, x, g1, N, C, H, W, 0, 1);
    wgrad1d_desc(wd[1], dt, t1, gn, N, C, H, W, 1, 1);
    wgrad1d_desc(wd[2], dt, t2, gb, N, C, H, W, 0, dil);
    wgrad1d_desc(wd[3], dt, t3, gs, N, C, H, W, 1, dil);
    return launch_wgrad_batch(wd, 4, dw, db, woff, 3, C * 3, 1u, ar, st);
}

// ================================================================================ Upsampler
static int upsampler_forward_impl(int dt, const void *x, const float *w, const float *b, const erf_bn_t *bn, void *y,
                                  void *pre, int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes,
                                  void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (Cin % 4 || Cout % 4) return fail("UpsamplerBlock: channel counts must be multiples of 4");
    const int mode = bn_mode(bn);
    const long long Mo = (long long)N * 4 * H * W;
    const bool two_pass = mode == 2 || (mode == 1 && pre);
    void *dst = two_pass ? pre : y;
    if (two_pass && !pre) {
        dst = ar.floats((size_t)Mo * Cout);
        if (!dst) return fail("UpsamplerBlock: workspace too small");
    }
    const ParityTaps pt = parity_taps();
    IgemmDesc d;
    zero_desc(d);
    d.dt = dt;
    d.in = x; d.w = w; d.out = dst;
    d.epi.bias = b;
    if (mode == 1 && !two_pass) d.epi.bn = bn_eval_affine(bn);
    d.epi.relu = !two_pass;
    d.N = N; d.GH = H; d.GW = W;
    d.in_H = H; d.in_W = W; d.in_C = Cin; d.K = Cin; d.ss = 1;
    d.out_H = 2 * H; d.out_W = 2 * W; d.out_C = Cout; d.Nout = Cout; d.os = 2;
    d.ngroups = 4;
    for (int g = 0; g < 4; ++g) {
        d.gtap0[g] = pt.gtap0[g]; d.gntaps[g] = pt.gntaps[g]; d.goa[g] = pt.goa[g]; d.gob[g] = pt.gob[g];
    }
    for (int t = 0; t < 9; ++t) { d.tdh[t] = pt.dh[t]; d.tdw[t] = pt.dw[t]; d.woff[t] = pt.kh[t] * 3 + pt.kw[t]; }
    d.wsk = Cout * 9; d.wsn = 9;       // W[ci][co][kh][kw]: k = ci, j = co
    if (up16_fwd_ok(d)) {              // 16 output channels: streaming kernel, all parity classes per wave
        if (launch_up16_fwd(d, st)) return 1;
    } else if (launch_igemm(d, ar, st)) return 1;
    if (two_pass) {
        if (mode == 2) { if (bn_train_stats(dt, bn, dst, Mo, Cout, ar, st)) return 1; }
        else if (bn_eval_prepare(bn, Cout, st)) return 1;
        Epilogue e = kNoEpi;
        e.bn = mode == 2 ? bn_saved_affine(bn) : bn_eval_affine(bn);
        e.relu = 1;
        if (launch_elt(dt, dst, y, e, Mo, Cout, 4 * H * W, st)) return 1;
    }
    return 0;
}

static int upsampler_backward_impl(int dt, const void *x, const float *w, const erf_bn_t *bn, const void *y,
                                   const void *pre, const void *gy, void *dx, float *dw, float *db,
                                   float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                                   void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    const int mode = bn_mode(bn);
    const long long Mo = (long long)N * 4 * H * W;
    const bool premasked = (flags & ERF_BWD_GY_PREMASKED) != 0;
    const void *g = gy;
    if (mode || !premasked) {
        float *gbuf = ar.floats((size_t)Mo * Cout);
        if (!gbuf) return fail("UpsamplerBlock backward: workspace too small");
        if (mode) {
            if (bn_backward(dt, bn, pre, gy, premasked ? nullptr : y, nullptr, gbuf, dgamma, dbeta, Mo, Cout,
                            4 * H * W, ar, st)) return 1;
        } else {
            if (launch_mask(dt, gy, y, gbuf, Mo, Cout, st)) return 1;
        }
        g = gbuf;
    }
    const ParityTaps pt = parity_taps();
    {   // weight gradient over the parity classes
        WgradDesc d;
        memset(&d, 0, sizeof(d));
        d.dt = dt;
        d.in = x; d.g = g;
        d.N = N; d.GH = H; d.GW = W;
        d.in_H = H; d.in_W = W; d.in_C = Cin; d.K = Cin; d.ss = 1; d.os = 2; d.ntaps = 9;
        int woff[9];
        for (int t = 0; t < 9; ++t) {
            d.tdh[t] = pt.dh[t]; d.tdw[t] = pt.dw[t]; d.toa[t] = pt.oa[t]; d.tob[t] = pt.ob[t];
            woff[t] = pt.kh[t] * 3 + pt.kw[t];
        }
        d.g_H = 2 * H; d.g_W = 2 * W; d.g_C = Cout; d.Nout = Cout;
        if (Cout == 16 && sampler16_shape_ok(Cin, (long long)N * H * W, Cin, Cout)) {
            if (launch_up16_wgrad(d, dw, db, Cout * 9, 9, 0, ar, st)) return 1;
        } else if (launch_wgrad(d, dw, db, woff, Cout * 9, 9, pt.rep_mask, ar, st)) return 1;
    }
    if (dx) {   // data gradient = ordinary 3x3 stride-2 convolution of g
        IgemmDesc d;
        zero_desc(d);
        d.dt = dt;
        d.in = g; d.w = w; d.out = dx;
        if (flags & ERF_BWD_MASK_DX) { d.epi.mask = x; d.epi.mask_C = Cin; }
        d.N = N; d.GH = H; d.GW = W;
        d.in_H = 2 * H; d.in_W = 2 * W; d.in_C = Cout; d.K = Cout; d.ss = 2;
        d.out_H = H; d.out_W = W; d.out_C = Cin; d.Nout = Cin;
        d.gntaps[0] = 9;
        for (int t = 0; t < 9; ++t) { d.tdh[t] = t / 3 - 1; d.tdw[t] = t % 3 - 1; d.woff[t] = t; }
        d.wsk = 9; d.wsn = Cout * 9;   // W[ci][co][kh][kw]: k = co, j = ci
        if (launch_igemm(d, ar, st)) return 1;
    }
    return 0;
}

// ================================================================================ output conv
// Blocks of the output-convolution kernels: they stride over the pixels, so a block fills its LDS
// weight tile once; about four resident blocks per CU.
static unsigned outconv_grid(long long M)
{
    const long long b = (M + 255) / 256;
    return (unsigned)(b < 1024 ? (b < 1 ? 1 : b) : 1024);
}

// dt = storage of x; out_dt = storage of the logits / probabilities
static int outconv_forward_impl(int dt, int out_dt, const void *x, const float *w, const float *b, void *y, int N,
                                int Cin, int H, int W, int ncls, int softmax, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (Cin != OC_CIN) return fail("output ConvTranspose2d: Cin must be %d (got %d)", OC_CIN, Cin);
    if (ncls < 1 || ncls > OC_MAX_NCLS) return fail("output ConvTranspose2d: 1 <= classes <= %d", OC_MAX_NCLS);
    if (!aligned16(x) || ((uintptr_t)y & (out_dt ? 3 : 7))) return fail("output ConvTranspose2d: misaligned tensors");
    const long long M = (long long)N * H * W;
    OutConvDesc d = {x, w, b, y, N, H, W, ncls};
    if (!dt && out_dt) return fail("output ConvTranspose2d: bf16 logits need bf16 activations");
    if (softmax) {
        if (out_dt) outconv_fwd_kernel<true, bf16_t, bf16_t><<<outconv_grid(M), 256, 0, st>>>(d);
        else BY_DT(dt, (outconv_fwd_kernel<true, T, float><<<outconv_grid(M), 256, 0, st>>>(d)));
    } else {
        if (out_dt) outconv_fwd_kernel<false, bf16_t, bf16_t><<<outconv_grid(M), 256, 0, st>>>(d);
        else BY_DT(dt, (outconv_fwd_kernel<false, T, float><<<outconv_grid(M), 256, 0, st>>>(d)));
    }
    CHECK_LAUNCH("outconv_fwd_kernel");
    return 0;
}

static int outconv_argmax_impl(int dt, const void *x, const float *w, const float *b, long long *labels, int N, int Cin,
                               int H, int W, int ncls, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (Cin != OC_CIN) return fail("output ConvTranspose2d: Cin must be %d (got %d)", OC_CIN, Cin);
    if (ncls < 1 || ncls > OC_MAX_NCLS) return fail("output ConvTranspose2d: 1 <= classes <= %d", OC_MAX_NCLS);
    if (!aligned16(x) || !labels) return fail("output ConvTranspose2d: misaligned tensors");
    const long long M = (long long)N * H * W;
    if (M == 0) return 0;
    OutConvDesc d = {x, w, b, nullptr, N, H, W, ncls};
    BY_DT(dt, (outconv_argmax_kernel<T><<<outconv_grid(M), 256, 0, st>>>(d, labels)));
    CHECK_LAUNCH("outconv_argmax_kernel");
    return 0;
}

static int outconv_backward_impl(int dt, const void *x, const float *w, const float *gy, void *dx, float *dw, float *db,
                                 int N, int Cin, int H, int W, int ncls, int flags, void *ws, size_t ws_bytes,
                                 void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (Cin != OC_CIN) return fail("output ConvTranspose2d: Cin must be %d (got %d)", OC_CIN, Cin);
    if (ncls < 1 || ncls > OC_MAX_NCLS) return fail("output ConvTranspose2d: 1 <= classes <= %d", OC_MAX_NCLS);
    const long long M = (long long)N * H * W;
    const int nab = ncls * 4, slab = nab * OC_CIN + nab;
    long long nsplit = (M + 1023) / 1024;
    if (nsplit > 512) nsplit = 512;
    long long sp = (M + nsplit - 1) / nsplit;
    sp = (sp + 15) / 16 * 16;
    nsplit = (M + sp - 1) / sp;
    if (M * OC_CIN >= (1LL << 29) || M * 4 * ncls >= (1LL << 29))
        return fail("output ConvTranspose2d backward: tensors of 2 GiB or more are not supported");
    float *partial = ar.floats((size_t)nsplit * slab);
    if (!partial) return fail("output ConvTranspose2d backward: workspace too small");
    OutConvBwdDesc d = {x, w, gy, dx, partial, N, H, W, ncls, (int)sp, (flags & ERF_BWD_MASK_DX) ? 1 : 0};
    if (dx) {
        BY_DT(dt, (outconv_dgrad_kernel<T><<<outconv_grid(M), 256, 0, st>>>(d)));
        CHECK_LAUNCH("outconv_dgrad_kernel");
    }
    BY_DT(dt, (outconv_wgrad_kernel<T><<<(unsigned)nsplit, 256, 0, st>>>(d)));
    CHECK_LAUNCH("outconv_wgrad_kernel");
    outconv_wreduce_kernel<<<(slab + 31) / 32, 256, 0, st>>>(partial, dw, db, (int)nsplit, ncls);
    CHECK_LAUNCH("outconv_wreduce_kernel");
    return 0;
}

// CrossEntropyLoss backward fused into the output transposed convolution's backward
static int outconv_ce_backward_impl(int dt, const void *x, const float *w, const float *b, const long long *target,
                                    const float *lse, const float *stats, const float *gy, void *dx, float *dw,
                                    float *db, int N, int Cin, int H, int W, int ncls, long long ignore_index,
                                    int flags, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (Cin != OC_CIN) return fail("output ConvTranspose2d: Cin must be %d (got %d)", OC_CIN, Cin);
    if (ncls < 1 || ncls > OC_MAX_NCLS) return fail("output ConvTranspose2d: 1 <= classes <= %d", OC_MAX_NCLS);
    if (!x || !w || !target || !lse || !stats || !gy || !dw) return fail("output ConvTranspose2d + loss backward: null pointer");
    if (!aligned16(x) || (dx && !aligned16(dx)) || ((uintptr_t)lse & 7) || ((uintptr_t)target & 7))
        return fail("output ConvTranspose2d + loss backward: misaligned tensors");
    const long long M = (long long)N * H * W;
    if (M == 0) return fail("output ConvTranspose2d + loss backward: empty input");
    if (M * OC_CIN >= (1LL << 29) || M * 4 * ncls >= (1LL << 29))
        return fail("output ConvTranspose2d backward: tensors of 2 GiB or more are not supported");
    const int nab = ncls * 4, slab = nab * OC_CIN + nab;
    const long long groups = (M + 15) / 16;
    long long nblk = 4LL * cu_count();
    long long gpb = (groups + nblk - 1) / nblk;
    gpb = (gpb + 3) / 4 * 4;                    // whole rounds of the four waves
    nblk = (groups + gpb - 1) / gpb;
    float *partial = ar.floats((size_t)nblk * slab);
    if (!partial) return fail("output ConvTranspose2d + loss backward: workspace too small");
    OutConvCeDesc d = {x, w, b, target, lse, stats, gy, dx, partial, N, H, W, ncls, (flags & ERF_BWD_MASK_DX) ? 1 : 0,
                       ignore_index, (unsigned)groups, (unsigned)gpb};
    if (ncls <= 20) BY_DT(dt, (outconv_ce_bwd_kernel<5, T><<<(unsigned)nblk, 256, 0, st>>>(d)));
    else BY_DT(dt, (outconv_ce_bwd_kernel<8, T><<<(unsigned)nblk, 256, 0, st>>>(d)));
    CHECK_LAUNCH("outconv_ce_bwd_kernel");
    outconv_wreduce_kernel<<<(slab + 31) / 32, 256, 0, st>>>(partial, dw, db, (int)nblk, ncls);
    CHECK_LAUNCH("outconv_wreduce_kernel");
    return 0;
}

// Output convolution + CrossEntropyLoss forward in one kernel (outconv_ce_fwd_kernel): loss, labelled
// pixel count, per-pixel log-sum-exp and (optionally) the accuracy counts, the logits never stored.
static int outconv_ce_forward_impl(int dt, const void *x, const float *w, const float *b, const long long *target,
                                   float *stats, float *lse, long long *counts, int N, int Cin, int H, int W, int ncls,
                                   long long ignore_index, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (Cin != OC_CIN) return fail("output ConvTranspose2d: Cin must be %d (got %d)", OC_CIN, Cin);
    if (ncls < 1 || ncls > OC_MAX_NCLS) return fail("output ConvTranspose2d: 1 <= classes <= %d", OC_MAX_NCLS);
    if (!x || !w || !target || !stats || !lse) return fail("output ConvTranspose2d + loss forward: null pointer");
    if (!aligned16(x) || !aligned16(target) || ((uintptr_t)lse & 7))
        return fail("output ConvTranspose2d + loss forward: misaligned tensors");
    const long long M = (long long)N * H * W;
    if (M == 0) return fail("output ConvTranspose2d + loss forward: empty input");
    if (M * OC_CIN >= (1LL << 29)) return fail("output ConvTranspose2d + loss forward: tensors of 2 GiB or more are not supported");
    const long long blocks = (M + 255) / 256;
    float *partial = ar.floats((size_t)blocks * 2);
    unsigned *cpartial = counts ? (unsigned *)ar.take((size_t)blocks * 2 * ncls * sizeof(unsigned)) : nullptr;
    float *wpk = ar.floats((size_t)ncls * 4 * OC_CIN + ncls);
    if (!partial || !wpk || (counts && !cpartial)) return fail("output ConvTranspose2d + loss forward: workspace too small");
    OutConvCeFwdDesc d = {x, w, b, wpk, target, lse, partial, cpartial, N, H, W, ncls, ignore_index};
    outconv_ce_pack_kernel<<<1, 256, 0, st>>>(d);
    CHECK_LAUNCH("outconv_ce_pack_kernel");
    if (ncls == 20) BY_DT(dt, (outconv_ce_fwd_kernel<T, 20><<<(unsigned)blocks, 256, 0, st>>>(d)));
    else if (ncls == 19) BY_DT(dt, (outconv_ce_fwd_kernel<T, 19><<<(unsigned)blocks, 256, 0, st>>>(d)));
    else BY_DT(dt, (outconv_ce_fwd_kernel<T, 0><<<(unsigned)blocks, 256, 0, st>>>(d)));
    CHECK_LAUNCH("outconv_ce_fwd_kernel");
    if (counts) {
        ce_reduce_counts_kernel<<<1 + 2 * ncls, 256, 0, st>>>(partial, (int)blocks, stats, cpartial, 2 * ncls, counts);
        CHECK_LAUNCH("ce_reduce_counts_kernel");
    } else {
        ce_reduce_kernel<<<1, 256, 0, st>>>(partial, (int)blocks, stats);
        CHECK_LAUNCH("ce_reduce_kernel");
    }
    return 0;
}

// ================================================================================ plain Conv2d
static void conv2d_taps(int k, int *dh, int *dw, int *woff, bool transpose)
{
    const int r = (k - 1) / 2;
    for (int t = 0; t < k * k; ++t) {
        const int a = t / k - r, b = t % k - r;
        dh[t] = transpose ? -a : a;
        dw[t] = transpose ? -b : b;
        woff[t] = t;
    }
}

static int conv2d_forward_impl(int dt, const void *x, const float *w, const float *b, void *y, int N, int Cin, int H,
                               int W, int Cout, int k, int relu, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (k != 1 && k != 3) return fail("conv2d: kernel size must be 1 or 3");
    IgemmDesc d;
    zero_desc(d);
    d.dt = dt;
    d.in = x; d.w = w; d.out = y;
    d.epi.bias = b; d.epi.relu = relu;
    d.N = N; d.GH = H; d.GW = W;
    d.in_H = H; d.in_W = W; d.in_C = Cin; d.K = Cin;
    d.out_H = H; d.out_W = W; d.out_C = Cout; d.Nout = Cout;
    d.gntaps[0] = k * k;
    conv2d_taps(k, d.tdh, d.tdw, d.woff, false);
    d.wsk = k * k; d.wsn = Cin * k * k;
    return launch_igemm(d, ar, st);
}

static int conv2d_backward_impl(int dt, const void *x, const float *w, const void *y, const void *gy, void *dx,
                                float *dw, float *db, int N, int Cin, int H, int W, int Cout, int k, int relu,
                                void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (k != 1 && k != 3) return fail("conv2d: kernel size must be 1 or 3");
    const long long M = (long long)N * H * W;
    const void *g = gy;
    if (relu) {
        if (Cout % 4) return fail("conv2d backward with ReLU needs Cout %% 4 == 0");
        float *gm = ar.floats((size_t)M * Cout);
        if (!gm) return fail("conv2d backward: workspace too small");
        if (launch_mask(dt, gy, y, gm, M, Cout, st)) return 1;
        g = gm;
    }
    WgradDesc wd;
    memset(&wd, 0, sizeof(wd));
    wd.dt = dt;
    wd.in = x; wd.g = g;
    wd.N = N; wd.GH = H; wd.GW = W;
    wd.in_H = H; wd.in_W = W; wd.in_C = Cin; wd.K = Cin; wd.ss = 1; wd.os = 1; wd.ntaps = k * k;
    int woff[9];
    conv2d_taps(k, wd.tdh, wd.tdw, woff, false);
    wd.g_H = H; wd.g_W = W; wd.g_C = Cout; wd.Nout = Cout;
    if (launch_wgrad(wd, dw, db, woff, k * k, Cin * k * k, 1u, ar, st)) return 1;
    if (dx) {
        IgemmDesc d;
        zero_desc(d);
        d.dt = dt;
        d.in = g; d.w = w; d.out = dx;
        d.N = N; d.GH = H; d.GW = W;
        d.in_H = H; d.in_W = W; d.in_C = Cout; d.K = Cout;
        d.out_H = H; d.out_W = W; d.out_C = Cin; d.Nout = Cin;
        d.gntaps[0] = k * k;
        conv2d_taps(k, d.tdh, d.tdw, d.woff, true);
        d.wsk = Cin * k * k; d.wsn = k * k;
        if (launch_igemm(d, ar, st)) return 1;
    }
    return 0;
}

// ================================================================================ CrossEntropyLoss2d
static int cross_entropy_forward_impl(const float *logits, const long long *target, float *stats, float *lse,
                                      long long *counts, int N, int K, int H, int W, long long ignore_index, void *ws,
                                      size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    {   // look the static scratch array's address up on this device while no capture can be under way
        // through this entry point's caller-provided workspace (erf_cross_entropy_forward_static then
        // never has to do it inside a capture); failure here is not an error of this call
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!static_scratch_resolved() && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
            void *unused = nullptr;
            if (static_scratch(&unused)) g_err[0] = 0;
        }
    }
    if (K < 1) return fail("cross entropy: no classes");
    const long long P = (long long)N * H * W;
    if (P == 0) return fail("cross entropy: empty input");
    const long long blocks = (P + CE_BLOCK - 1) / CE_BLOCK;
    if (blocks >= (1LL << 31)) return fail("cross entropy: too many pixels");
    if (counts && K > CE_MAX_K) return fail("cross entropy with accuracy counts: at most %d classes (got %d)", CE_MAX_K, K);
    float *partial = ar.floats((size_t)blocks * 2);
    unsigned *cpartial = counts ? (unsigned *)ar.take((size_t)blocks * 2 * K * sizeof(unsigned)) : nullptr;
    if (!partial || (counts && !cpartial)) return fail("cross entropy: workspace too small");
    CeDesc d = {logits, target, lse, partial, P, H * W, K, ignore_index, cpartial};
    if (K <= CE_MAX_K) ce_fwd_kernel<true><<<(unsigned)blocks, CE_BLOCK, 0, st>>>(d);
    else ce_fwd_kernel<false><<<(unsigned)blocks, CE_BLOCK, 0, st>>>(d);
    CHECK_LAUNCH("ce_fwd_kernel");
    if (counts) {
        ce_reduce_counts_kernel<<<1 + 2 * K, 256, 0, st>>>(partial, (int)blocks, stats, cpartial, 2 * K, counts);
        CHECK_LAUNCH("ce_reduce_counts_kernel");
    } else {
        ce_reduce_kernel<<<1, 256, 0, st>>>(partial, (int)blocks, stats);
        CHECK_LAUNCH("ce_reduce_kernel");
    }
    return 0;
}

int erf_cross_entropy_forward(const float *logits, const long long *target, float *stats, float *lse,
                              int N, int K, int H, int W, long long ignore_index, void *ws,
                              size_t ws_bytes, void *stream)
{
    return cross_entropy_forward_impl(logits, target, stats, lse, nullptr, N, K, H, W, ignore_index, ws, ws_bytes,
                                      stream);
}

int erf_cross_entropy_forward_counts(const float *logits, const long long *target, float *stats, float *lse,
                                     long long *counts, int N, int K, int H, int W, long long ignore_index,
                                     void *ws, size_t ws_bytes, void *stream)
{
    if (!counts) return fail("cross entropy with accuracy counts: null pointer");
    return cross_entropy_forward_impl(logits, target, stats, lse, counts, N, K, H, W, ignore_index, ws, ws_bytes,
                                      stream);
}

// ---- the same loss for callers that may not allocate (static scratch, see the top of the file)
int erf_cross_entropy_forward_static(const float *logits, const long long *target, float *stats, float *lse,
                                     int N, int K, int H, int W, long long ignore_index, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (K < 1) return fail("cross entropy: no classes");
    if (N < 0 || H < 0 || W < 0) return fail("cross entropy: negative extent");
    const long long P = (long long)N * H * W;
    if (P == 0) return fail("cross entropy: empty input");
    const long long blocks = (P + CE_BLOCK - 1) / CE_BLOCK;
    const long long max_blocks = ERF_CE_STATIC_BYTES / (2 * sizeof(float));
    if (blocks > max_blocks)
        return fail("cross entropy (static scratch): %lld pixels exceed the limit of %lld", P,
                    max_blocks * CE_BLOCK);
    if (!logits || !target || !stats || !lse) return fail("cross entropy: null pointer");
    void *sp = nullptr;
    if (!static_scratch_resolved()) {
        // the first call on a device looks the array's address up; that is not a stream operation and
        // must not fall inside a capture (every erf_cross_entropy_forward* call does the look-up too)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail("cross entropy (static scratch): the first call on a device must be made outside stream "
                        "capture (it resolves the scratch array's address)");
    }
    if (static_scratch(&sp)) return 1;
    // every block overwrites its own pair of partial sums before the reduction reads them, so
    // nothing of an earlier call's contents has to be cleared
    CeDesc d = {logits, target, lse, (float *)sp, P, H * W, K, ignore_index, nullptr};
    if (K <= CE_MAX_K) ce_fwd_kernel<true><<<(unsigned)blocks, CE_BLOCK, 0, st>>>(d);
    else ce_fwd_kernel<false><<<(unsigned)blocks, CE_BLOCK, 0, st>>>(d);
    CHECK_LAUNCH("ce_fwd_kernel");
    ce_reduce_kernel<<<1, 256, 0, st>>>((const float *)sp, (int)blocks, stats);
    CHECK_LAUNCH("ce_reduce_kernel");
    return 0;
}

int erf_cross_entropy_backward(const float *logits, const long long *target, const float *lse,
                               const float *stats, const float *gy, float *dlogits, int N, int K, int H,
                               int W, long long ignore_index, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const long long P = (long long)N * H * W;
    if (P == 0 || K < 1) return fail("cross entropy: empty input");
    const long long blocks = (P + CE_BLOCK - 1) / CE_BLOCK;
    CeBwdDesc d = {logits, target, lse, stats, gy, dlogits, P, H * W, K, ignore_index};
    ce_bwd_kernel<<<(unsigned)blocks, CE_BLOCK, 0, st>>>(d);
    CHECK_LAUNCH("ce_bwd_kernel");
    return 0;
}

// ================================================================================ masked regression losses
static int ml_check(const char *what, int kind, const float *a, const float *b, const float *c, const float *target,
                    int N, int K, int H, int W, long long &P, long long &blocks)
{
    if (kind < ML_L1 || kind > ML_LLCP) return fail("%s: unknown kind %d", what, kind);
    if (N < 0 || H < 0 || W < 0) return fail("%s: negative extent", what);
    const bool cls = kind == ML_L1CP || kind == ML_LLCP;
    if (cls ? (K < 1 || K > ML_MAX_K) : K != 1)
        return fail("%s: kind %d takes %s (got %d)", what, kind, cls ? "1..256 classes" : "one value per pixel", K);
    P = (long long)N * H * W;
    if (P == 0) return fail("%s: empty input", what);
    if ((long long)H * W >= (1LL << 31)) return fail("%s: image too large", what);
    blocks = (P + ML_BLOCK * ML_PPT - 1) / (ML_BLOCK * ML_PPT);
    if (blocks >= (1LL << 31)) return fail("%s: too many pixels", what);
    if (!a || !target || (kind != ML_L1 && !b) || (kind == ML_LLCP && !c)) return fail("%s: null pointer", what);
    return 0;
}

int erf_masked_loss_forward(int kind, const float *a, const float *b, const float *c, const float *target,
                            float eps, float *stats, int N, int K, int H, int W, void *ws, size_t ws_bytes,
                            void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    long long P, blocks;
    if (ml_check("masked loss", kind, a, b, c, target, N, K, H, W, P, blocks)) return 1;
    if (!stats) return fail("masked loss: null pointer");
    MlDesc d;
    memset(&d, 0, sizeof(d));
    d.a = a; d.b = b; d.c = c; d.target = target; d.P = P; d.HW = H * W; d.K = K; d.eps = eps;
    d.partial = ar.floats((size_t)blocks * 2);
    if (!d.partial) return fail("masked loss: workspace too small");
    switch (kind) {
    case ML_L1: ml_fwd_kernel<ML_L1><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    case ML_LL: ml_fwd_kernel<ML_LL><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    case ML_L1CP: ml_fwd_kernel<ML_L1CP><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    default: ml_fwd_kernel<ML_LLCP><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    }
    CHECK_LAUNCH("ml_fwd_kernel");
    ce_reduce_kernel<<<1, 256, 0, st>>>(d.partial, (int)blocks, stats);
    CHECK_LAUNCH("ce_reduce_kernel");
    return 0;
}

int erf_masked_loss_backward(int kind, const float *a, const float *b, const float *c, const float *target,
                             float eps, const float *stats, const float *gy, float *da, float *db, float *dc,
                             int N, int K, int H, int W, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    long long P, blocks;
    if (ml_check("masked loss backward", kind, a, b, c, target, N, K, H, W, P, blocks)) return 1;
    if (!stats || !gy || !da || (kind == ML_LL && !db)) return fail("masked loss backward: null pointer");
    MlDesc d;
    memset(&d, 0, sizeof(d));
    d.a = a; d.b = b; d.c = c; d.target = target; d.P = P; d.HW = H * W; d.K = K; d.eps = eps;
    d.stats = stats; d.gy = gy; d.da = da; d.db = db;
    const bool cls = kind == ML_L1CP || kind == ML_LLCP;
    if (cls) {
        d.partial = ar.floats((size_t)blocks * 2 * K);
        if (!d.partial) return fail("masked loss backward: workspace too small");
    }
    switch (kind) {
    case ML_L1: ml_bwd_kernel<ML_L1><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    case ML_LL: ml_bwd_kernel<ML_LL><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    case ML_L1CP: ml_bwd_kernel<ML_L1CP><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    default: ml_bwd_kernel<ML_LLCP><<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d); break;
    }
    CHECK_LAUNCH("ml_bwd_kernel");
    if (cls && (db || dc)) {
        ml_colsum_kernel<<<(unsigned)K, 256, 0, st>>>(d.partial, (int)blocks, K, db, kind == ML_LLCP ? dc : nullptr);
        CHECK_LAUNCH("ml_colsum_kernel");
    }
    return 0;
}

int erf_mse_mean_forward(const float *a, const float *b, long long n, float *stats, void *ws, size_t ws_bytes,
                         void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (n <= 0) return fail("mse: empty input");
    if (!a || !b || !stats) return fail("mse: null pointer");
    const long long blocks = (n + ML_BLOCK * ML_PPT - 1) / (ML_BLOCK * ML_PPT);
    if (blocks >= (1LL << 31)) return fail("mse: too many elements");
    MseDesc d;
    memset(&d, 0, sizeof(d));
    d.a = a; d.b = b; d.n = n;
    d.partial = ar.floats((size_t)blocks * 2);
    if (!d.partial) return fail("mse: workspace too small");
    mse_fwd_kernel<<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d);
    CHECK_LAUNCH("mse_fwd_kernel");
    ce_reduce_kernel<<<1, 256, 0, st>>>(d.partial, (int)blocks, stats);
    CHECK_LAUNCH("ce_reduce_kernel");
    return 0;
}

int erf_mse_mean_backward(const float *a, const float *b, const float *gy, float *da, long long n, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) return fail("mse: empty input");
    if (!a || !b || !gy || !da) return fail("mse: null pointer");
    const long long blocks = (n + ML_BLOCK * ML_PPT - 1) / (ML_BLOCK * ML_PPT);
    if (blocks >= (1LL << 31)) return fail("mse: too many elements");
    MseDesc d;
    memset(&d, 0, sizeof(d));
    d.a = a; d.b = b; d.gy = gy; d.da = da; d.n = n;
    d.scale = (float)(1.0 / (double)n);
    mse_bwd_kernel<<<(unsigned)blocks, ML_BLOCK, 0, st>>>(d);
    CHECK_LAUNCH("mse_bwd_kernel");
    return 0;
}

// ================================================================================ parameter updates
int erf_adam_step(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                  float *const *exp_avg_sq, const long long *numel, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void *stream)
{
    if (step < 1) return fail("adam: step counts from 1 (got %lld)", step);
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0)) return fail("adam: betas outside [0, 1)");
    if (!(eps >= 0.0) || !(lr >= 0.0) || !(weight_decay >= 0.0)) return fail("adam: negative lr / eps / weight decay");
    OptDesc d;
    memset(&d, 0, sizeof(d));
    // the scalars in double, as torch.optim.Adam forms them on the host, rounded once for the kernel
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    d.c0 = (float)(1.0 - beta1); d.c1 = (float)beta2; d.c2 = (float)(1.0 - beta2);
    d.c3 = (float)(lr / bc1); d.c4 = (float)sqrt(bc2); d.c5 = (float)eps;
    d.wd = (float)weight_decay;
    return opt_launch<OPT_ADAM>("adam", count, params, grads, exp_avg, exp_avg_sq, numel, d, (hipStream_t)stream);
}

int erf_sgd_step(int count, float *const *params, const float *const *grads, float *const *momentum_buf,
                 const long long *numel, double lr, double momentum, double weight_decay, void *stream)
{
    if (!(lr >= 0.0) || !(momentum >= 0.0) || !(weight_decay >= 0.0)) return fail("sgd: negative lr / momentum / weight decay");
    OptDesc d;
    memset(&d, 0, sizeof(d));
    d.c0 = (float)momentum; d.c3 = (float)lr; d.wd = (float)weight_decay;
    return opt_launch<OPT_SGD>("sgd", count, params, grads, momentum_buf, nullptr, numel, d, (hipStream_t)stream);
}

int erf_ema_update(int count, float *const *target, const float *const *source, const long long *numel,
                   double discount, void *stream)
{
    OptDesc d;
    memset(&d, 0, sizeof(d));
    d.c0 = (float)(1.0 - discount); d.c1 = (float)discount;
    return opt_launch<OPT_EMA>("ema", count, target, source, nullptr, nullptr, numel, d, (hipStream_t)stream);
}

// ================================================================================ iouEval counts
int erf_iou_add_batch(const long long *pred, const long long *target, long long n_pixels, int n_classes,
                      int ignore_index, long long *counts, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (n_classes < 1 || n_classes > IOU_MAX_CLASSES) return fail("iouEval: 1..%d classes (got %d)", IOU_MAX_CLASSES, n_classes);
    if (ignore_index >= n_classes || ignore_index < -1) return fail("iouEval: ignore index %d outside [-1, %d)", ignore_index, n_classes);
    if (!pred || !target || !counts) return fail("iouEval: null pointer");
    if (n_pixels <= 0) return 0;
    IouDesc d;
    d.pred = pred; d.target = target; d.n = n_pixels;
    d.n_classes = n_classes;
    d.counted = ignore_index < 0 ? n_classes : ignore_index;     // reference slices [:, :ignoreIndex]
    d.ignore = ignore_index;
    d.counts = counts;
    if (d.counted == 0) return 0;
    long long blocks = (n_pixels + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    d.blocks = (int)blocks;
    d.partial = (unsigned *)ar.take((size_t)blocks * 3 * d.counted * sizeof(unsigned));
    if (!d.partial) return fail("iouEval: workspace too small");
    iou_count_kernel<<<(unsigned)blocks, 256, 0, st>>>(d);
    CHECK_LAUNCH("iou_count_kernel");
    iou_sum_kernel<<<1, 256, 0, st>>>(d);
    CHECK_LAUNCH("iou_sum_kernel");
    return 0;
}

int erf_accuracy_counts(const float *scores, const long long *target, int N, int K, int H, int W,
                        long long *counts, void *ws, size_t ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    if (K < 1 || K > IOU_MAX_CLASSES) return fail("accuracy: 1..%d classes (got %d)", IOU_MAX_CLASSES, K);
    if (!scores || !target || !counts) return fail("accuracy: null pointer");
    const long long P = (long long)N * H * W;
    if (P <= 0) return 0;
    AccDesc d;
    d.scores = scores; d.target = target; d.P = P; d.HW = H * W; d.K = K; d.counts = counts;
    long long blocks = (P + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    d.blocks = (int)blocks;
    d.partial = (unsigned *)ar.take((size_t)blocks * 2 * K * sizeof(unsigned));
    if (!d.partial) return fail("accuracy: workspace too small");
    acc_count_kernel<<<(unsigned)blocks, 256, 0, st>>>(d);
    CHECK_LAUNCH("acc_count_kernel");
    acc_sum_kernel<<<1, 256, 0, st>>>(d);
    CHECK_LAUNCH("acc_sum_kernel");
    return 0;
}


// ================================================================================ public entry points
// Each block call exists for fp32 tensors (the reference's arithmetic) and, suffixed _bf16, for bf16
// activation tensors (fp32 parameters, fp32 accumulation); both run the same host logic above.
#define ERF_DT_F32 0
#define ERF_DT_BF16 1

int erf_downsampler_forward(const float *x, long long sn, long long sc, long long sh, long long sw,
                            const float *w, const float *b, const erf_bn_t *bn, float *y, float *pre,
                            int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes, void *stream)
{
    return downsampler_forward_impl(ERF_DT_F32, x, sn, sc, sh, sw, w, b, bn, y, pre, N, Cin, H, W, Cout, ws, ws_bytes, stream);
}
int erf_downsampler_forward_bf16(const void *x, long long sn, long long sc, long long sh, long long sw,
                                 const float *w, const float *b, const erf_bn_t *bn, void *y, void *pre,
                                 int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes, void *stream)
{
    return downsampler_forward_impl(ERF_DT_BF16, x, sn, sc, sh, sw, w, b, bn, y, pre, N, Cin, H, W, Cout, ws, ws_bytes, stream);
}
int erf_downsampler_backward(const float *x, long long sn, long long sc, long long sh, long long sw,
                             const float *w, const erf_bn_t *bn, const float *y, const float *pre,
                             const float *gy, float *dx, float *dw, float *db, float *dgamma,
                             float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                             void *ws, size_t ws_bytes, void *stream)
{
    return downsampler_backward_impl(ERF_DT_F32, x, sn, sc, sh, sw, w, bn, y, pre, gy, dx, dw, db, dgamma, dbeta, N, Cin,
                                     H, W, Cout, flags, ws, ws_bytes, stream);
}
int erf_downsampler_backward_bf16(const void *x, long long sn, long long sc, long long sh, long long sw,
                                  const float *w, const erf_bn_t *bn, const void *y, const void *pre,
                                  const void *gy, void *dx, float *dw, float *db, float *dgamma,
                                  float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                                  void *ws, size_t ws_bytes, void *stream)
{
    return downsampler_backward_impl(ERF_DT_BF16, x, sn, sc, sh, sw, w, bn, y, pre, gy, dx, dw, db, dgamma, dbeta, N, Cin,
                                     H, W, Cout, flags, ws, ws_bytes, stream);
}

int erf_nb1d_forward(const float *x, const float *const w[4], const float *const b[4],
                     const erf_bn_t *bn1, const erf_bn_t *bn2, const float *drop_scale, int dil, float *y,
                     float *const save[5], int N, int C, int H, int W, void *ws, size_t ws_bytes,
                     void *stream)
{
    return nb1d_forward_impl(ERF_DT_F32, x, w, b, bn1, bn2, drop_scale, dil, y, (void *const *)save, N, C, H, W, ws,
                             ws_bytes, stream);
}
int erf_nb1d_forward_bf16(const void *x, const float *const w[4], const float *const b[4],
                          const erf_bn_t *bn1, const erf_bn_t *bn2, const float *drop_scale, int dil, void *y,
                          void *const save[5], int N, int C, int H, int W, void *ws, size_t ws_bytes,
                          void *stream)
{
    return nb1d_forward_impl(ERF_DT_BF16, x, w, b, bn1, bn2, drop_scale, dil, y, save, N, C, H, W, ws, ws_bytes, stream);
}
int erf_nb1d_chain_forward(const float *x, const erf_nb1d_block_t *blocks, int nblocks, int N, int C,
                           int H, int W, void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_chain_forward_impl(ERF_DT_F32, x, blocks, nblocks, N, C, H, W, ws, ws_bytes, stream);
}
int erf_nb1d_chain_forward_bf16(const void *x, const erf_nb1d_block_t *blocks, int nblocks, int N, int C,
                                int H, int W, void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_chain_forward_impl(ERF_DT_BF16, x, blocks, nblocks, N, C, H, W, ws, ws_bytes, stream);
}
int erf_nb1d_backward(const float *x, const float *const w[4], const erf_bn_t *bn1, const erf_bn_t *bn2,
                      const float *drop_scale, int dil, const float *y, const float *const save[5],
                      const float *gy, float *dx, float *const dw[4], float *const db[4],
                      float *const dgamma[2], float *const dbeta[2], int N, int C, int H, int W, int flags,
                      void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_backward_impl(ERF_DT_F32, x, w, bn1, bn2, drop_scale, dil, y, (const void *const *)save, nullptr, gy, dx,
                              dw, db, dgamma, dbeta, N, C, H, W, flags, ws, ws_bytes, stream);
}
int erf_nb1d_backward_bits(const float *x, const float *const w[4], const erf_bn_t *bn1, const erf_bn_t *bn2,
                           const float *drop_scale, int dil, const float *y, const float *const save[5],
                           const unsigned *const sign_bits[4], const float *gy, float *dx, float *const dw[4],
                           float *const db[4], float *const dgamma[2], float *const dbeta[2], int N, int C, int H,
                           int W, int flags, void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_backward_impl(ERF_DT_F32, x, w, bn1, bn2, drop_scale, dil, y, (const void *const *)save, sign_bits, gy,
                              dx, dw, db, dgamma, dbeta, N, C, H, W, flags, ws, ws_bytes, stream);
}
int erf_nb1d_backward_bits_bf16(const void *x, const float *const w[4], const erf_bn_t *bn1, const erf_bn_t *bn2,
                                const float *drop_scale, int dil, const void *y, const void *const save[5],
                                const unsigned *const sign_bits[4], const void *gy, void *dx, float *const dw[4],
                                float *const db[4], float *const dgamma[2], float *const dbeta[2], int N, int C, int H,
                                int W, int flags, void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_backward_impl(ERF_DT_BF16, x, w, bn1, bn2, drop_scale, dil, y, save, sign_bits, gy, dx, dw, db, dgamma,
                              dbeta, N, C, H, W, flags, ws, ws_bytes, stream);
}
int erf_nb1d_backward_bf16(const void *x, const float *const w[4], const erf_bn_t *bn1, const erf_bn_t *bn2,
                           const float *drop_scale, int dil, const void *y, const void *const save[5],
                           const void *gy, void *dx, float *const dw[4], float *const db[4],
                           float *const dgamma[2], float *const dbeta[2], int N, int C, int H, int W, int flags,
                           void *ws, size_t ws_bytes, void *stream)
{
    return nb1d_backward_impl(ERF_DT_BF16, x, w, bn1, bn2, drop_scale, dil, y, save, nullptr, gy, dx, dw, db, dgamma,
                              dbeta, N, C, H, W, flags, ws, ws_bytes, stream);
}

int erf_upsampler_forward(const float *x, const float *w, const float *b, const erf_bn_t *bn, float *y,
                          float *pre, int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes,
                          void *stream)
{
    return upsampler_forward_impl(ERF_DT_F32, x, w, b, bn, y, pre, N, Cin, H, W, Cout, ws, ws_bytes, stream);
}
int erf_upsampler_forward_bf16(const void *x, const float *w, const float *b, const erf_bn_t *bn, void *y,
                               void *pre, int N, int Cin, int H, int W, int Cout, void *ws, size_t ws_bytes,
                               void *stream)
{
    return upsampler_forward_impl(ERF_DT_BF16, x, w, b, bn, y, pre, N, Cin, H, W, Cout, ws, ws_bytes, stream);
}
int erf_upsampler_backward(const float *x, const float *w, const erf_bn_t *bn, const float *y,
                           const float *pre, const float *gy, float *dx, float *dw, float *db,
                           float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                           void *ws, size_t ws_bytes, void *stream)
{
    return upsampler_backward_impl(ERF_DT_F32, x, w, bn, y, pre, gy, dx, dw, db, dgamma, dbeta, N, Cin, H, W, Cout, flags,
                                   ws, ws_bytes, stream);
}
int erf_upsampler_backward_bf16(const void *x, const float *w, const erf_bn_t *bn, const void *y,
                                const void *pre, const void *gy, void *dx, float *dw, float *db,
                                float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout, int flags,
                                void *ws, size_t ws_bytes, void *stream)
{
    return upsampler_backward_impl(ERF_DT_BF16, x, w, bn, y, pre, gy, dx, dw, db, dgamma, dbeta, N, Cin, H, W, Cout, flags,
                                   ws, ws_bytes, stream);
}

int erf_outconv_forward(const float *x, const float *w, const float *b, float *y, int N, int Cin, int H,
                        int W, int ncls, int softmax, void *stream)
{
    return outconv_forward_impl(ERF_DT_F32, ERF_DT_F32, x, w, b, y, N, Cin, H, W, ncls, softmax, stream);
}
int erf_outconv_forward_bf16(const void *x, const float *w, const float *b, void *y, int y_is_bf16, int N,
                             int Cin, int H, int W, int ncls, int softmax, void *stream)
{
    return outconv_forward_impl(ERF_DT_BF16, y_is_bf16 ? ERF_DT_BF16 : ERF_DT_F32, x, w, b, y, N, Cin, H, W, ncls,
                                softmax, stream);
}
int erf_outconv_ce_backward(const float *x, const float *w, const float *b, const long long *target,
                            const float *lse, const float *stats, const float *gy, float *dx, float *dw, float *db,
                            int N, int Cin, int H, int W, int ncls, long long ignore_index, int flags, void *ws,
                            size_t ws_bytes, void *stream)
{
    return outconv_ce_backward_impl(ERF_DT_F32, x, w, b, target, lse, stats, gy, dx, dw, db, N, Cin, H, W, ncls,
                                    ignore_index, flags, ws, ws_bytes, stream);
}
int erf_outconv_ce_backward_bf16(const void *x, const float *w, const float *b, const long long *target,
                                 const float *lse, const float *stats, const float *gy, void *dx, float *dw,
                                 float *db, int N, int Cin, int H, int W, int ncls, long long ignore_index, int flags,
                                 void *ws, size_t ws_bytes, void *stream)
{
    return outconv_ce_backward_impl(ERF_DT_BF16, x, w, b, target, lse, stats, gy, dx, dw, db, N, Cin, H, W, ncls,
                                    ignore_index, flags, ws, ws_bytes, stream);
}
int erf_outconv_ce_forward(const float *x, const float *w, const float *b, const long long *target, float *stats,
                           float *lse, long long *counts, int N, int Cin, int H, int W, int ncls,
                           long long ignore_index, void *ws, size_t ws_bytes, void *stream)
{
    return outconv_ce_forward_impl(ERF_DT_F32, x, w, b, target, stats, lse, counts, N, Cin, H, W, ncls, ignore_index,
                                   ws, ws_bytes, stream);
}
int erf_outconv_ce_forward_bf16(const void *x, const float *w, const float *b, const long long *target, float *stats,
                                float *lse, long long *counts, int N, int Cin, int H, int W, int ncls,
                                long long ignore_index, void *ws, size_t ws_bytes, void *stream)
{
    return outconv_ce_forward_impl(ERF_DT_BF16, x, w, b, target, stats, lse, counts, N, Cin, H, W, ncls, ignore_index,
                                   ws, ws_bytes, stream);
}
int erf_outconv_argmax(const float *x, const float *w, const float *b, long long *labels, int N, int Cin,
                       int H, int W, int ncls, void *stream)
{
    return outconv_argmax_impl(ERF_DT_F32, x, w, b, labels, N, Cin, H, W, ncls, stream);
}
int erf_outconv_argmax_bf16(const void *x, const float *w, const float *b, long long *labels, int N, int Cin,
                            int H, int W, int ncls, void *stream)
{
    return outconv_argmax_impl(ERF_DT_BF16, x, w, b, labels, N, Cin, H, W, ncls, stream);
}
int erf_outconv_backward(const float *x, const float *w, const float *gy, float *dx, float *dw, float *db,
                         int N, int Cin, int H, int W, int ncls, int flags, void *ws, size_t ws_bytes,
                         void *stream)
{
    return outconv_backward_impl(ERF_DT_F32, x, w, gy, dx, dw, db, N, Cin, H, W, ncls, flags, ws, ws_bytes, stream);
}
int erf_outconv_backward_bf16(const void *x, const float *w, const float *gy, void *dx, float *dw, float *db,
                              int N, int Cin, int H, int W, int ncls, int flags, void *ws, size_t ws_bytes,
                              void *stream)
{
    return outconv_backward_impl(ERF_DT_BF16, x, w, gy, dx, dw, db, N, Cin, H, W, ncls, flags, ws, ws_bytes, stream);
}

int erf_conv2d_forward(const float *x, const float *w, const float *b, float *y, int N, int Cin, int H,
                       int W, int Cout, int k, int relu, void *ws, size_t ws_bytes, void *stream)
{
    return conv2d_forward_impl(ERF_DT_F32, x, w, b, y, N, Cin, H, W, Cout, k, relu, ws, ws_bytes, stream);
}
int erf_conv2d_forward_bf16(const void *x, const float *w, const float *b, void *y, int N, int Cin, int H,
                            int W, int Cout, int k, int relu, void *ws, size_t ws_bytes, void *stream)
{
    return conv2d_forward_impl(ERF_DT_BF16, x, w, b, y, N, Cin, H, W, Cout, k, relu, ws, ws_bytes, stream);
}
int erf_conv2d_backward(const float *x, const float *w, const float *y, const float *gy, float *dx,
                        float *dw, float *db, int N, int Cin, int H, int W, int Cout, int k, int relu,
                        void *ws, size_t ws_bytes, void *stream)
{
    return conv2d_backward_impl(ERF_DT_F32, x, w, y, gy, dx, dw, db, N, Cin, H, W, Cout, k, relu, ws, ws_bytes, stream);
}
int erf_conv2d_backward_bf16(const void *x, const float *w, const void *y, const void *gy, void *dx,
                             float *dw, float *db, int N, int Cin, int H, int W, int Cout, int k, int relu,
                             void *ws, size_t ws_bytes, void *stream)
{
    return conv2d_backward_impl(ERF_DT_BF16, x, w, y, gy, dx, dw, db, N, Cin, H, W, Cout, k, relu, ws, ws_bytes, stream);
}

}  // extern "C"
