// Probability map -> detections on gfx950.  Replaces TextDetector._post_process
// (app/ml/models/text_detector.py:143-178): strict threshold, cv2.findContours(RETR_EXTERNAL), contourArea >= 100,
// minAreaRect -> boxPoints -> np.int0, bbox clamp/scale/size filter, mean-probability confidence.
//
// The reference walks contours one after another on the CPU.  Here every frame of the batch is
// processed at once and nothing is traced:
//   * external contours  = 8-connected components of the complement of the frame-connected background
//                          (4-connected, image virtually zero padded): one lock-free union-find over all
//                          pixels (atomicMin label equivalence) finds the frame background, a second pass
//                          joins foreground, holes and nested islands into the filled components
//   * contourArea        = lattice identity on the filled component: #2x2 blocks fully inside + 1/2 #blocks
//                          with exactly three pixels inside (equals the shoelace area of the traced border)
//   * minAreaRect        = per-row min/max x of the component -> two monotone chains -> strict hull ->
//                          rotating calipers in float32, replayed in the published operation order
//                          (this file is compiled with -ffp-contract=off)
//   * ordering           = components are numbered by their raster-first pixel with block scans, the
//                          surviving detections are emitted in reverse raster order (cv2 4.8.1's order)
// All stages are HBM/L2-bound integer work; no host synchronisation happens between them.
#include <float.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#include "../../include/vtd.h"
#include "vtd_common.h"

namespace {

constexpr int SCAN_THREADS = 1024;
constexpr double VTD_PI = 3.1415926535897932384626433832795;

struct PostWs {
    const float* prob;  // [n, h, w]
    int n, h, w, P;
    float thr;
    int* label;        // [n][P+1], index 0 = frame background sentinel, pixel i <-> i+1
    uint64_t* fgbits;  // [n][h][wpr] foreground, 1 bit per pixel
    uint64_t* inbits;  // [n][h][wpr] inside (foreground or hole: not frame background)
    int wpr;           // 64-pixel words per row
    int* compid;       // [n][P]  valid at component roots
    int* ncomp;        // [n]
    int* slice_count;  // [n][h*wpr] roots per word, then their exclusive prefix
    int maxc;          // capacity of the per-component arrays
    int* area2;        // [n][maxc]  2 * contour area
    int* bbox;         // [n][maxc][4] xmin,xmax,ymin,ymax
    int* rowoff;       // [n][maxc]  -1 for non-candidates
    int* candlist;     // [n][maxcand]
    int* ncand;        // [n]
    int maxcand;
    int* rowmin;       // [n][P]
    int* rowmax;       // [n][P]
    vtd_detection* cand_rec;  // [n][maxcand]
    int* cand_valid;   // [n][maxcand]
    const int* orig_w; // [n]
    const int* orig_h; // [n]
    vtd_detection* out; // [n][max_out]
    int* out_count;    // [n]
    int max_out;
};

__device__ __forceinline__ int ld_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// find with path halving.  Vertical structures (the background strip beside a text box, a tall stroke) are joined row
// by row and would otherwise leave parent chains as long as the structure is tall for every later lookup.  A halving
// store only ever replaces a non-root's parent by one of its ancestors (same set, smaller index), so it is safe next to
// concurrent atomicMin unions: a link it overwrites was already re-united by the thread that lost it.
__device__ __forceinline__ int uf_find(int* L, int a) {
    int p = ld_relaxed(L + a);
    while (p != a) {
        const int gp = ld_relaxed(L + p);
        if (gp != p) __hip_atomic_store(L + a, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = p;
        p = gp;
    }
    return a;
}

// Lock-free union by minimum index: parents only ever decrease, so stale reads are harmless and the
// atomicMin at the device-coherent point decides.
__device__ __forceinline__ void uf_unite(int* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;
    }
}

// The per-pixel stages work on bit-packed rows: one 64-bit word = 64 consecutive pixels of a row (`wpr` words per row),
// one thread per word.  Neighbour tests are shifts/ANDs of the word, the row above and their edge carries; only the set
// bits of the resulting masks (run contacts, class changes) touch the union-find.
struct Words3 { uint64_t c, l, r; };
__device__ __forceinline__ Words3 load3(const uint64_t* B, int y, int seg, int h, int wpr) {
    Words3 o{0, 0, 0};
    if (y < 0 || y >= h) return o;
    const uint64_t* row = B + (int64_t)y * wpr;
    o.c = row[seg];
    o.l = seg > 0 ? row[seg - 1] : 0;
    o.r = seg + 1 < wpr ? row[seg + 1] : 0;
    return o;
}
__device__ __forceinline__ uint64_t west(const Words3& v) { return (v.c << 1) | (v.l >> 63); }   // bit i = pixel i-1
__device__ __forceinline__ uint64_t east(const Words3& v) { return (v.c >> 1) | (v.r << 63); }   // bit i = pixel i+1
__device__ __forceinline__ uint64_t valid_mask(int seg, int w, int wpr) {
    return (seg == wpr - 1 && (w & 63)) ? ((1ull << (w & 63)) - 1ull) : ~0ull;
}
// lowest maximal run of ones in m: returns its mask, start in *i0, length in *len
__device__ __forceinline__ uint64_t first_run(uint64_t m, int* i0, int* len) {
    const int i = __ffsll((long long)m) - 1;
    const uint64_t t = m >> i;
    const int n = (~t == 0) ? 64 - i : __ffsll((long long)~t) - 1;
    *i0 = i;
    *len = n;
    return (n == 64 ? ~0ull : ((1ull << n) - 1ull)) << i;
}

// ---- stage 1: threshold + horizontal runs.  One wave owns one row and walks its words: a ballot of the foreground bit
// is the packed word and gives every lane the start of its same-class run, carried across word seams, which becomes its
// initial union-find parent.  Horizontal connectivity therefore costs no atomics at all, and the (usually huge) frame
// background collapses to one run per row instead of one node per pixel.
__device__ __forceinline__ void pp_init_row(const PostWs& ws, int f, int y, int lane) {
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    int carry_start = 0;      // x of the start of the run that reaches the previous word's last pixel
    bool carry_fg = false;
    const float* prow = ws.prob + (int64_t)f * ws.P + (int64_t)y * ws.w;
    // the row's words are fetched 16 at a time before any of them is processed: one memory round trip per 1024 pixels instead of one
    // per word (the per-frame kernel below has 16 waves for a whole map: nobody else covers a dependent chain of round trips)
    for (int s0 = 0; s0 < ws.wpr; s0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = (s0 + k) * 64 + lane;
            v[k] = (s0 + k < ws.wpr && x < ws.w) ? prow[x] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int seg = s0 + k;
            if (seg >= ws.wpr) continue;   // (wave-uniform)
            const int x = seg * 64 + lane;
            const bool valid = x < ws.w;
            const int pix = y * ws.w + x;
            const bool fg = valid && (v[k] > ws.thr);
            const unsigned long long m = __ballot(fg);
            const unsigned long long same = fg ? m : ~m;
            const unsigned long long below = lane ? (~same & ((1ull << lane) - 1ull)) : 0ull;
            int start = below ? seg * 64 + 64 - __clzll((long long)below) : seg * 64;
            if (!below && seg > 0 && fg == carry_fg) start = carry_start;  // the run continues from the previous word
            if (valid) {
                int parent = y * ws.w + start + 1;
                // background on the frame's border belongs to the (virtual) outside: hang it on the sentinel
                if (!fg && (y == 0 || y == ws.h - 1 || x == 0)) parent = 0;
                L[pix + 1] = parent;
            }
            carry_start = __shfl(start, 63);
            carry_fg = (m >> 63) & 1;
            if (lane == 0) ws.fgbits[((int64_t)f * ws.h + y) * ws.wpr + seg] = m;
        }
    }
    if (y == 0 && lane == 0) L[0] = 0;
}

__global__ __launch_bounds__(256) void pp_init(const PostWs ws) {
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)ws.n * ws.h;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ri = wave0; ri < total; ri += nwaves) {
        const int f = (int)(ri / ws.h), y = (int)(ri - (int64_t)f * ws.h);
        pp_init_row(ws, f, y, lane);
    }
}

#define PP_FOR_EACH_WORD(f, y, seg, wi)                                                                            \
    const int64_t total_words = (int64_t)ws.n * ws.h * ws.wpr;                                                     \
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < total_words; wi += (int64_t)gridDim.x * blockDim.x)

// ---- stage 2: join runs.  Only the first column of every run-to-run contact issues a union.
__device__ __forceinline__ void pp_merge_fg_bg_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    const uint64_t* B = ws.fgbits + (int64_t)f * ws.h * ws.wpr;
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const uint64_t vm = valid_mask(seg, ws.w, ws.wpr);
    const Words3 cur = load3(B, y, seg, ws.h, ws.wpr), up = load3(B, y - 1, seg, ws.h, ws.wpr);
    const int base = y * ws.w + seg * 64 + 1;  // label index of bit 0
    if (seg == ws.wpr - 1 && !((cur.c >> ((ws.w - 1) & 63)) & 1)) uf_unite(L, base + ((ws.w - 1) & 63), 0);  // background at the right frame edge
    if (y == 0) return;
    const uint64_t W = west(cur), E = east(cur), N = up.c, NW = west(up), NE = east(up);
    // foreground, 8-connected
    uint64_t m = cur.c & N & ~(W & NW);
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w); }
    m = cur.c & ~N & NW & ~W;
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w - 1); }
    m = cur.c & ~N & NE & ~E;
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w + 1); }
    // background, 4-connected (pixels left of the frame do not exist: bit 0 of word 0 has no west neighbour)
    const uint64_t bg = ~cur.c & vm, bgN = ~up.c & vm;
    const uint64_t bgW = (~cur.c << 1) | (seg > 0 ? (~cur.l) >> 63 : 0), bgNW = (~up.c << 1) | (seg > 0 ? (~up.l) >> 63 : 0);
    m = bg & bgN & ~(bgW & bgNW);
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w); }
}

__global__ void pp_merge_fg_bg(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_merge_fg_bg_word(ws, f, y, seg, wi);
    }
}

// ---- stage 3a: background runs that do not reach the frame background are holes -> inside mask
__device__ __forceinline__ void pp_classify_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const uint64_t cur = ws.fgbits[wi];
    uint64_t bg = ~cur & valid_mask(seg, ws.w, ws.wpr), holes = 0;
    const int base = y * ws.w + seg * 64 + 1;
    while (bg) {
        int i0, len;
        const uint64_t run = first_run(bg, &i0, &len);
        if (uf_find(L, base + i0) != 0) holes |= run;
        bg &= ~run;
    }
    ws.inbits[wi] = cur | holes;
}

__global__ void pp_classify(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_classify_word(ws, f, y, seg, wi);
    }
}

// ---- stage 3b: filled components = foreground + holes + islands.  Same-class neighbours are already joined (diagonal
// hole pixels always share a 4-connected or foreground bridge), so only class changes need unions.
__device__ __forceinline__ void pp_merge_inside_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    const uint64_t* F = ws.fgbits + (int64_t)f * ws.h * ws.wpr;
    const uint64_t* I = ws.inbits + (int64_t)f * ws.h * ws.wpr;
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const Words3 fc = load3(F, y, seg, ws.h, ws.wpr), fu = load3(F, y - 1, seg, ws.h, ws.wpr);
    const Words3 ic = load3(I, y, seg, ws.h, ws.wpr), iu = load3(I, y - 1, seg, ws.h, ws.wpr);
    if (!ic.c) return;
    const int base = y * ws.w + seg * 64 + 1;
    uint64_t m = ic.c & west(ic) & (fc.c ^ west(fc));
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - 1); }
    m = ic.c & iu.c & (fc.c ^ fu.c);
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w); }
    m = ic.c & west(iu) & (fc.c ^ west(fu));
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w - 1); }
    m = ic.c & east(iu) & (fc.c ^ east(fu));
    while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(L, base + i, base + i - ws.w + 1); }
}

__global__ void pp_merge_inside(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_merge_inside_word(ws, f, y, seg, wi);
    }
}

// ---- stage 4: component roots.  A root is the raster-first pixel of its filled component, hence a foreground run start;
// only those are examined (and path-compressed), counted per word, scanned per frame and numbered in raster order.
__device__ __forceinline__ uint64_t fg_run_starts(uint64_t cur) { return cur & ~(cur << 1); }

__device__ __forceinline__ void pp_count_roots_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const int base = y * ws.w + seg * 64 + 1;
    uint64_t m = fg_run_starts(ws.fgbits[wi]);
    int cnt = 0;
    while (m) {
        const int i = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int r = uf_find(L, base + i);
        L[base + i] = r;
        cnt += r == base + i;
    }
    ws.slice_count[wi] = cnt;
}

__global__ void pp_count_roots(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_count_roots_word(ws, f, y, seg, wi);
    }
}

// exclusive scan of one int per thread across a 1024-thread block; returns the exclusive prefix, *total = sum
__device__ int block_exclusive_scan(int v, int* sh /* [2*SCAN_THREADS] */, int* total) {
    const int t = threadIdx.x;
    int* a = sh;
    int* b = sh + SCAN_THREADS;
    a[t] = v;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
        b[t] = a[t] + (t >= d ? a[t - d] : 0);
        __syncthreads();
        int* tmp = a; a = b; b = tmp;
    }
    const int incl = a[t];
    *total = a[SCAN_THREADS - 1];
    __syncthreads();
    return incl - v;
}

// one block per frame: exclusive scan of the per-word root counts (raster order), reset the per-component accumulators
__device__ void pp_scan_slices_frame(const PostWs& ws, int f, int* sh /* [2 * SCAN_THREADS] */) {
    const int slices = ws.h * ws.wpr;
    int* cnt = ws.slice_count + (int64_t)f * slices;
    const int chunk = (slices + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(threadIdx.x * chunk, slices), hi = min(lo + chunk, slices);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    int total;
    int base = block_exclusive_scan(sum, sh, &total);
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        cnt[i] = base;
        base += c;
    }
    if (total > ws.maxc) total = ws.maxc;  // cannot happen: maxc bounds the number of 8-connected components
    if (threadIdx.x == 0) ws.ncomp[f] = total;
    for (int c = threadIdx.x; c < total; c += SCAN_THREADS) {
        ws.area2[(int64_t)f * ws.maxc + c] = 0;
        int* bb = ws.bbox + ((int64_t)f * ws.maxc + c) * 4;
        bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void pp_scan_slices(const PostWs ws) {
    __shared__ int sh[2 * SCAN_THREADS];
    pp_scan_slices_frame(ws, blockIdx.x, sh);
}

__device__ __forceinline__ void pp_number_components_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    const int* L = ws.label + (int64_t)f * (ws.P + 1);
    const int base = y * ws.w + seg * 64 + 1;
    uint64_t m = fg_run_starts(ws.fgbits[wi]);
    int next = ws.slice_count[wi];
    while (m) {
        const int i = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (L[base + i] == base + i) ws.compid[(int64_t)f * ws.P + base + i - 1] = next++;
    }
}

__global__ void pp_number_components(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_number_components_word(ws, f, y, seg, wi);
    }
}

// ---- stage 5: per-component contour area (lattice identity) and bounding box, one set of atomics per inside run.
// 2x2 blocks are anchored at their top-left pixel and credited to the run that owns the block's first inside pixel of the
// top row (every block with >= 3 inside pixels has one).
__device__ __forceinline__ void pp_stats_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    const uint64_t* I = ws.inbits + (int64_t)f * ws.h * ws.wpr;
    const Words3 a = load3(I, y, seg, ws.h, ws.wpr);
    if (!a.c) return;
    const Words3 b = load3(I, y + 1, seg, ws.h, ws.wpr);
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const uint64_t a1 = east(a), b1 = east(b);
    const uint64_t four = a.c & a1 & b.c & b1;
    const uint64_t three = (a.c & a1 & b.c & ~b1) | (a.c & a1 & ~b.c & b1) | (a.c & ~a1 & b.c & b1) | (~a.c & a1 & b.c & b1);
    // block anchored on the last pixel of the previous word whose only missing corner is that pixel: owned by bit 0 here
    const int carry = (seg > 0 && !(a.l >> 63) && (a.c & 1) && (b.l >> 63) && (b.c & 1)) ? 1 : 0;
    const int base = y * ws.w + seg * 64 + 1;
    uint64_t m = a.c;
    while (m) {
        int i0, len;
        const uint64_t run = first_run(m, &i0, &len);
        m &= ~run;
        const uint64_t left = i0 > 0 ? 1ull << (i0 - 1) : 0ull;
        const int add = 2 * __popcll(four & run) + __popcll(three & (run | left)) + (i0 == 0 ? carry : 0);
        const int c = ws.compid[(int64_t)f * ws.P + uf_find(L, base + i0) - 1];
        if (add) atomicAdd(ws.area2 + (int64_t)f * ws.maxc + c, add);
        int* bb = ws.bbox + ((int64_t)f * ws.maxc + c) * 4;
        atomicMin(bb + 0, seg * 64 + i0); atomicMax(bb + 1, seg * 64 + i0 + len - 1); atomicMin(bb + 2, y); atomicMax(bb + 3, y);
    }
}

__global__ void pp_stats(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_stats_word(ws, f, y, seg, wi);
    }
}

// one block per frame: candidates (contourArea >= 100) in raster order + their row-table offsets
__device__ void pp_candidates_frame(const PostWs& ws, int f, int* sh /* [2 * SCAN_THREADS] */) {
    const int nc = ws.ncomp[f];
    const int* area2 = ws.area2 + (int64_t)f * ws.maxc;
    const int* bbox = ws.bbox + (int64_t)f * ws.maxc * 4;
    int* rowoff = ws.rowoff + (int64_t)f * ws.maxc;
    const int chunk = (nc + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(threadIdx.x * chunk, nc), hi = min(lo + chunk, nc);
    int cnt = 0, rows = 0;
    for (int c = lo; c < hi; ++c)
        if (area2[c] >= 200) { cnt++; rows += bbox[4 * c + 3] - bbox[4 * c + 2] + 1; }
    int tot_c, tot_r;
    int base_c = block_exclusive_scan(cnt, sh, &tot_c);
    int base_r = block_exclusive_scan(rows, sh, &tot_r);
    for (int c = lo; c < hi; ++c) {
        if (area2[c] >= 200 && base_c < ws.maxcand) {
            ws.candlist[(int64_t)f * ws.maxcand + base_c] = c;
            rowoff[c] = base_r;
            base_c++;
            const int rows_c = bbox[4 * c + 3] - bbox[4 * c + 2] + 1;
            for (int r = 0; r < rows_c; ++r) {
                ws.rowmin[(int64_t)f * ws.P + base_r + r] = 0x7fffffff;
                ws.rowmax[(int64_t)f * ws.P + base_r + r] = -1;
            }
            base_r += rows_c;
        } else {
            rowoff[c] = -1;
        }
    }
    if (threadIdx.x == 0) ws.ncand[f] = min(tot_c, ws.maxcand);
}

__global__ __launch_bounds__(SCAN_THREADS) void pp_candidates(const PostWs ws) {
    __shared__ int sh[2 * SCAN_THREADS];
    pp_candidates_frame(ws, blockIdx.x, sh);
}

__device__ __forceinline__ void pp_row_extents_word(const PostWs& ws, int f, int y, int seg, int64_t wi) {
    int* L = ws.label + (int64_t)f * (ws.P + 1);
    const int base = y * ws.w + seg * 64 + 1;
    uint64_t m = ws.inbits[wi];
    while (m) {
        int i0, len;
        const uint64_t run = first_run(m, &i0, &len);
        m &= ~run;
        const int c = ws.compid[(int64_t)f * ws.P + uf_find(L, base + i0) - 1];
        const int off = ws.rowoff[(int64_t)f * ws.maxc + c];
        if (off < 0) continue;
        const int r = off + y - ws.bbox[((int64_t)f * ws.maxc + c) * 4 + 2];
        atomicMin(ws.rowmin + (int64_t)f * ws.P + r, seg * 64 + i0);
        atomicMax(ws.rowmax + (int64_t)f * ws.P + r, seg * 64 + i0 + len - 1);
    }
}

__global__ void pp_row_extents(const PostWs ws) {
    PP_FOR_EACH_WORD(f, y, seg, wi) {
        const int f = (int)(wi / ((int64_t)ws.h * ws.wpr));
        const int rem = (int)(wi - (int64_t)f * ws.h * ws.wpr);
        const int y = rem / ws.wpr, seg = rem - y * ws.wpr;
        pp_row_extents_word(ws, f, y, seg, wi);
    }
}

// ---- stages 1-6 of the chain above as ONE launch: a 1024-thread workgroup per frame.  The stages are loops over the frame's rows /
// 64-pixel words with two per-frame scans in between, so the ten kernel boundaries become workgroup barriers -- the union-find still lives
// in global memory (1.6 MB of labels per 640 x 640 map), its atomics still execute in L2.  What a boundary did besides ordering was
// drop the CU's L1: a word a stage read (or initialised) with a plain access and another wave then changed by an L2 atomic must not be
// served from that L1 to the next stage, so every stage change is barrier -> one wave invalidates the L1 (agent-scope acquire) ->
// barrier.  Why: the ten launches were 0.3 ms of summed in-situ kernel time per batch, each spread thin over every CU the detector's and
// the recogniser's wide kernels want (round-3 review); here a batch's post-process holds 16 waves on n CUs, once.  An OPTION
// (VTD_PP_FUSED=1), not the default: it measured 1 % slower end to end (vtd_postproc_run below).
__device__ __forceinline__ void pp_stage_sync() {
    __syncthreads();   // every wave's stores and atomics of the stage have been issued and acknowledged (s_waitcnt vmcnt(0) in front of the barrier)
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

__global__ __launch_bounds__(SCAN_THREADS) void pp_frame_kernel(const PostWs ws) {
    __shared__ int sh[2 * SCAN_THREADS];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int words = ws.h * ws.wpr;
    const int64_t w0 = (int64_t)f * words;
    for (int y = tid >> 6; y < ws.h; y += SCAN_THREADS / 64) pp_init_row(ws, f, y, lane);
    pp_stage_sync();
#define PP_FRAME_STAGE(fn)                                                    \
    for (int i = tid; i < words; i += SCAN_THREADS) {                         \
        const int y = i / ws.wpr, seg = i - y * ws.wpr;                       \
        fn(ws, f, y, seg, w0 + i);                                            \
    }                                                                         \
    pp_stage_sync();
    PP_FRAME_STAGE(pp_merge_fg_bg_word)
    PP_FRAME_STAGE(pp_classify_word)
    PP_FRAME_STAGE(pp_merge_inside_word)
    PP_FRAME_STAGE(pp_count_roots_word)
    pp_scan_slices_frame(ws, f, sh);
    pp_stage_sync();
    PP_FRAME_STAGE(pp_number_components_word)
    PP_FRAME_STAGE(pp_stats_word)
    pp_candidates_frame(ws, f, sh);
    pp_stage_sync();
    for (int i = tid; i < words; i += SCAN_THREADS) {
        const int y = i / ws.wpr, seg = i - y * ws.wpr;
        pp_row_extents_word(ws, f, y, seg, w0 + i);
    }
#undef PP_FRAME_STAGE
}

struct fpt { float x, y; };

// rotating calipers, minimum-area rectangle (float32, published operation order); out[6].  vect / inv_len are
// precomputed by the whole workgroup; this part is the inherently sequential walk and runs on one lane out of LDS.
__device__ void min_area_rect_dev(const fpt* points, int n, const fpt* vect, const float* inv_len, float* out) {
    float minarea = FLT_MAX;
    int left = 0, bottom = 0, right = 0, top = 0;
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    left_x = right_x = points[0].x;
    top_y = bottom_y = points[0].y;
    for (int i = 1; i < n; i++) {
        const fpt pt0 = points[i];
        if (pt0.x < left_x) { left_x = pt0.x; left = i; }
        if (pt0.x > right_x) { right_x = pt0.x; right = i; }
        if (pt0.y > top_y) { top_y = pt0.y; top = i; }
        if (pt0.y < bottom_y) { bottom_y = pt0.y; bottom = i; }
    }
    {
        double ax = vect[n - 1].x, ay = vect[n - 1].y;
        for (int i = 0; i < n; i++) {
            const double bx = vect[i].x, by = vect[i].y;
            const double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    int s0 = bottom, s1 = right, s2 = top, s3 = left;  // calipers contact points (scalars: no private-array indexing)
    int best_left = 0, best_bottom = 0;
    float best_a = 0, best_b = 0, best_w = 0, best_h = 0;
    for (int k = 0; k < n; k++) {
        const fpt v0 = vect[s0], v1 = vect[s1], v2 = vect[s2], v3 = vect[s3];
        const float dp0 = +base_a * v0.x + base_b * v0.y;
        const float dp1 = -base_b * v1.x + base_a * v1.y;
        const float dp2 = -base_a * v2.x - base_b * v2.y;
        const float dp3 = +base_b * v3.x - base_a * v3.y;
        float maxcos = dp0 * inv_len[s0];
        int main_element = 0;
        float cosalpha = dp1 * inv_len[s1];
        if (cosalpha > maxcos) { main_element = 1; maxcos = cosalpha; }
        cosalpha = dp2 * inv_len[s2];
        if (cosalpha > maxcos) { main_element = 2; maxcos = cosalpha; }
        cosalpha = dp3 * inv_len[s3];
        if (cosalpha > maxcos) { main_element = 3; maxcos = cosalpha; }
        {
            const int pindex = main_element == 0 ? s0 : main_element == 1 ? s1 : main_element == 2 ? s2 : s3;
            const float lead_x = vect[pindex].x * inv_len[pindex];
            const float lead_y = vect[pindex].y * inv_len[pindex];
            switch (main_element) {
                case 0: base_a = lead_x; base_b = lead_y; s0 = s0 + 1 == n ? 0 : s0 + 1; break;
                case 1: base_a = lead_y; base_b = -lead_x; s1 = s1 + 1 == n ? 0 : s1 + 1; break;
                case 2: base_a = -lead_x; base_b = -lead_y; s2 = s2 + 1 == n ? 0 : s2 + 1; break;
                default: base_a = -lead_y; base_b = lead_x; s3 = s3 + 1 == n ? 0 : s3 + 1; break;
            }
        }
        {
            float dx = points[s1].x - points[s3].x;
            float dy = points[s1].y - points[s3].y;
            const float width = dx * base_a + dy * base_b;
            dx = points[s2].x - points[s0].x;
            dy = points[s2].y - points[s0].y;
            const float height = -dx * base_b + dy * base_a;
            const float area = width * height;
            if (area <= minarea) {
                minarea = area;
                best_left = s3; best_a = base_a; best_w = width;
                best_b = base_b; best_h = height; best_bottom = s0;
            }
        }
    }
    {
        const float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
        const float C1 = A1 * points[best_left].x + points[best_left].y * B1;
        const float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
        const float idet = 1.f / (A1 * B2 - A2 * B1);
        const float px = (C1 * B2 - C2 * B1) * idet;
        const float py = (A1 * C2 - A2 * C1) * idet;
        out[0] = px; out[1] = py;
        out[2] = A1 * best_w; out[3] = B1 * best_w;
        out[4] = A2 * best_h; out[5] = B2 * best_h;
    }
}

// One 256-thread workgroup per candidate.
//   1. the candidate's row table (min / max x per row) is staged into LDS
//   2. hull membership is decided for every row in parallel: row r's right end is a vertex of the clockwise hull iff
//      min_{i<r} slope(i,r) > max_{j>r} slope(r,j) (x as a function of the row, exact integer fraction compares), its
//      left end iff max_{i<r} slope < min_{j>r} slope -- the same strict hull the sequential monotone chain builds
//   3. one lane compacts the vertices into the published order (clockwise on screen, ending at the raster-first pixel),
//      all lanes compute the edge vectors / inverse lengths, one lane walks the calipers
//   4. all lanes average the probability slice of the resulting box
constexpr int BOX_THREADS = 256;
__device__ __forceinline__ bool frac_lt(int an, int ad, int bn, int bd) { return (long long)an * bd < (long long)bn * ad; }  // an/ad < bn/bd, ad,bd > 0

__global__ __launch_bounds__(BOX_THREADS) void pp_boxes(const PostWs ws) {
    extern __shared__ __attribute__((aligned(16))) int box_lds[];
    const int f = blockIdx.y;
    __shared__ int sh_i[8];
    __shared__ double sh_acc[BOX_THREADS / 64];
    const int hcap = ws.h;
    int* rmin = box_lds;                       // [h]
    int* rmax = box_lds + hcap;                // [h]
    int* isv = box_lds + 2 * hcap;             // [h] bit0: right end is a hull vertex, bit1: left end is
    fpt* pts = (fpt*)(box_lds + 3 * hcap);     // [2h+2]
    fpt* vect = pts + 2 * hcap + 2;            // [2h+2]
    float* inv_len = (float*)(vect + 2 * hcap + 2);  // [2h+2]
  for (int k = blockIdx.x; k < ws.ncand[f]; k += gridDim.x) {
    __syncthreads();
    const int c = ws.candlist[(int64_t)f * ws.maxcand + k];
    const int* bb = ws.bbox + ((int64_t)f * ws.maxc + c) * 4;
    const int ytop = bb[2], H = bb[3] - bb[2] + 1;
    const int off = ws.rowoff[(int64_t)f * ws.maxc + c];
    for (int r = threadIdx.x; r < H; r += BOX_THREADS) {
        rmin[r] = ws.rowmin[(int64_t)f * ws.P + off + r];
        rmax[r] = ws.rowmax[(int64_t)f * ws.P + off + r];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < H; r += BOX_THREADS) {
        int flag = 0;
        if (r == 0 || r == H - 1) {
            flag = 3;
        } else {
            // right ends: concave envelope of rmax(row)
            int mn_n = rmax[r] - rmax[0], mn_d = r;             // min over i<r of (x_r - x_i)/(r - i)
            int mnL_n = rmin[r] - rmin[0], mnL_d = r;           // max over i<r for the left ends
            for (int i = 1; i < r; ++i) {
                const int n1 = rmax[r] - rmax[i], n2 = rmin[r] - rmin[i], d = r - i;
                if (frac_lt(n1, d, mn_n, mn_d)) { mn_n = n1; mn_d = d; }
                if (frac_lt(mnL_n, mnL_d, n2, d)) { mnL_n = n2; mnL_d = d; }
            }
            int mx_n = rmax[r + 1] - rmax[r], mx_d = 1;         // max over j>r of (x_j - x_r)/(j - r)
            int mxL_n = rmin[r + 1] - rmin[r], mxL_d = 1;       // min over j>r for the left ends
            for (int j = r + 2; j < H; ++j) {
                const int n1 = rmax[j] - rmax[r], n2 = rmin[j] - rmin[r], d = j - r;
                if (frac_lt(mx_n, mx_d, n1, d)) { mx_n = n1; mx_d = d; }
                if (frac_lt(n2, d, mxL_n, mxL_d)) { mxL_n = n2; mxL_d = d; }
            }
            if (frac_lt(mx_n, mx_d, mn_n, mn_d)) flag |= 1;     // strictly convex turn on the right chain
            if (frac_lt(mnL_n, mnL_d, mxL_n, mxL_d)) flag |= 2; // and on the left chain
        }
        isv[r] = flag;
    }
    __syncthreads();
    vtd_detection rec;
    if (threadIdx.x == 0) {
        // clockwise on screen starting just after the raster-first pixel S = (rmin[0], ytop) and ending at S:
        // right ends going down, then left ends going up (corners shared by both chains are emitted once)
        int n = 0;
        for (int r = 0; r < H; ++r)
            if ((isv[r] & 1) && !(r == 0 && rmax[0] == rmin[0])) { pts[n].x = (float)rmax[r]; pts[n].y = (float)(ytop + r); ++n; }
        for (int r = H - 1; r >= 0; --r)
            if ((isv[r] & 2) && !(r == H - 1 && rmin[r] == rmax[r])) { pts[n].x = (float)rmin[r]; pts[n].y = (float)(ytop + r); ++n; }
        sh_i[5] = n;
    }
    __syncthreads();
    const int n = sh_i[5];
    for (int i = threadIdx.x; i < n; i += BOX_THREADS) {
        const fpt p0 = pts[i], p1 = pts[i + 1 < n ? i + 1 : 0];
        const double dx = (double)p1.x - (double)p0.x, dy = (double)p1.y - (double)p0.y;
        vect[i].x = (float)dx;
        vect[i].y = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float cx = 0, cy = 0, bw = 0, bh = 0, angle = 0;
        if (n > 2) {
            float o[6];
            min_area_rect_dev(pts, n, vect, inv_len, o);
            cx = o[0] + (o[2] + o[4]) * 0.5f;
            cy = o[1] + (o[3] + o[5]) * 0.5f;
            bw = (float)sqrt((double)o[2] * o[2] + (double)o[3] * o[3]);
            bh = (float)sqrt((double)o[4] * o[4] + (double)o[5] * o[5]);
            angle = (float)atan2((double)o[3], (double)o[2]);
        } else if (n == 2) {
            cx = (pts[0].x + pts[1].x) * 0.5f;
            cy = (pts[0].y + pts[1].y) * 0.5f;
            const double dx = (double)pts[1].x - pts[0].x, dy = (double)pts[1].y - pts[0].y;
            bw = (float)sqrt(dx * dx + dy * dy);
            angle = (float)atan2(dy, dx);
        } else if (n == 1) {
            cx = pts[0].x; cy = pts[0].y;
        }
        angle = (float)((double)(angle * 180.f) / VTD_PI);
        const double rad = (double)angle * VTD_PI / 180.;
        const float b = (float)cos(rad) * 0.5f;
        const float a = (float)sin(rad) * 0.5f;
        float p[8];
        p[0] = cx - a * bh - b * bw;
        p[1] = cy + b * bh - a * bw;
        p[2] = cx + a * bh - b * bw;
        p[3] = cy - b * bh - a * bw;
        p[4] = 2 * cx - p[0];
        p[5] = 2 * cy - p[1];
        p[6] = 2 * cx - p[2];
        p[7] = 2 * cy - p[3];
        int xmin = 0x7fffffff, xmax = -0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
        for (int i = 0; i < 4; ++i) {
            const int xi = (int)p[2 * i], yi = (int)p[2 * i + 1];  // np.int0: truncate toward zero
            rec.polygon[2 * i] = xi; rec.polygon[2 * i + 1] = yi;
            xmin = min(xmin, xi); xmax = max(xmax, xi); ymin = min(ymin, yi); ymax = max(ymax, yi);
        }
        const long long ow = ws.orig_w[f], oh = ws.orig_h[f];
        // text_detector.py:160-166 -- clamp to the literal 640, int(v * W / 640) (exact as integer division for v >= 0)
        const long long x1 = (long long)max(0, xmin) * ow / 640, y1 = (long long)max(0, ymin) * oh / 640;
        const long long x2 = (long long)min(640, xmax) * ow / 640, y2 = (long long)min(640, ymax) * oh / 640;
        rec.bbox[0] = (int)x1; rec.bbox[1] = (int)y1; rec.bbox[2] = (int)x2; rec.bbox[3] = (int)y2;
        const int valid = (x2 - x1 > 10 && y2 - y1 > 10) ? 1 : 0;
        // text_detector.py:169-170 slice bounds with numpy clamping
        sh_i[0] = valid;
        sh_i[1] = (int)min((long long)ws.h, y1 * 640 / oh);
        sh_i[2] = (int)min((long long)ws.h, y2 * 640 / oh);
        sh_i[3] = (int)min((long long)ws.w, x1 * 640 / ow);
        sh_i[4] = (int)min((long long)ws.w, x2 * 640 / ow);
        rec.area = 0.5f * (float)ws.area2[(int64_t)f * ws.maxc + c];
        rec.first_x = rmin[0]; rec.first_y = ytop;
    }
    __syncthreads();
    const int valid = sh_i[0];
    if (!valid) {
        if (threadIdx.x == 0) ws.cand_valid[(int64_t)f * ws.maxcand + k] = 0;
        continue;
    }
    const int sy0 = sh_i[1], sy1 = sh_i[2], sx0 = sh_i[3], sx1 = sh_i[4];
    const int64_t cnt = (int64_t)max(sx1 - sx0, 0) * max(sy1 - sy0, 0);
    double acc = 0.0;
    const float* P = ws.prob + (int64_t)f * ws.P;
    for (int yy = sy0 + (threadIdx.x >> 5); yy < sy1; yy += BOX_THREADS / 32)   // 8 rows x 32 columns per pass
        for (int xx = sx0 + (threadIdx.x & 31); xx < sx1; xx += 32) acc += (double)P[(int64_t)yy * ws.w + xx];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) sh_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = 0.0;
        for (int i = 0; i < BOX_THREADS / 64; ++i) acc += sh_acc[i];
        rec.confidence = cnt > 0 ? (float)(acc / (double)cnt) : __builtin_nanf("");
        ws.cand_rec[(int64_t)f * ws.maxcand + k] = rec;
        ws.cand_valid[(int64_t)f * ws.maxcand + k] = 1;
    }
  }
}

// one block per frame: compact the surviving candidates, reverse raster order
__global__ __launch_bounds__(SCAN_THREADS) void pp_emit(const PostWs ws) {
    __shared__ int sh[2 * SCAN_THREADS];
    const int f = blockIdx.x;
    const int nc = ws.ncand[f];
    const int* valid = ws.cand_valid + (int64_t)f * ws.maxcand;
    const int chunk = (nc + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(threadIdx.x * chunk, nc), hi = min(lo + chunk, nc);
    int cnt = 0;
    for (int k = lo; k < hi; ++k) cnt += valid[k];
    int total;
    int base = block_exclusive_scan(cnt, sh, &total);
    for (int k = lo; k < hi; ++k)
        if (valid[k]) {
            const int dst = total - 1 - base;
            if (dst < ws.max_out) ws.out[(int64_t)f * ws.max_out + dst] = ws.cand_rec[(int64_t)f * ws.maxcand + k];
            base++;
        }
    if (threadIdx.x == 0) ws.out_count[f] = total;
}

}  // namespace

struct vtd_postproc {
    int max_batch = 0, h = 0, w = 0, max_out = 0;
    PostWs ws;
    void* blocks[24];
    int nblocks = 0;
    int *orig_w_dev = nullptr, *orig_h_dev = nullptr;
    std::vector<int> orig_shadow;  // host copy of (w, h) per frame as last uploaded
};

static int pp_alloc(vtd_postproc* pp, void** out, size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (bytes + 255) & ~size_t(255));
    if (e != hipSuccess) return -(int)e;
    pp->blocks[pp->nblocks++] = p;
    *out = p;
    return 0;
}

extern "C" {

int vtd_postproc_create(int max_batch, int map_h, int map_w, int max_out, vtd_postproc** out) {
    if (!out || max_batch <= 0 || map_h <= 0 || map_w <= 0 || max_out <= 0 || (int64_t)map_h * map_w > (1 << 26)) return -1100;
    if (map_h > 2900 || map_w >= 32768) return -1100;  // the per-candidate hull workspace lives in LDS; coordinates must keep fraction compares in range
    auto* pp = new vtd_postproc();
    pp->max_batch = max_batch; pp->h = map_h; pp->w = map_w; pp->max_out = max_out;
    PostWs& ws = pp->ws;
    const int64_t P = (int64_t)map_h * map_w;
    ws.h = map_h; ws.w = map_w; ws.P = (int)P;
    ws.maxc = ((map_h + 1) / 2) * ((map_w + 1) / 2) + 1;
    ws.maxcand = (int)(P / 100) + 1;
    ws.max_out = max_out;
    const int64_t B = max_batch;
    int rc = 0;
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.label, B * (P + 1) * 4);
    ws.wpr = (map_w + 63) / 64;
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.fgbits, B * map_h * ws.wpr * 8);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.inbits, B * map_h * ws.wpr * 8);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.compid, B * P * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.ncomp, B * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.slice_count, B * map_h * ws.wpr * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.area2, B * ws.maxc * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.bbox, B * ws.maxc * 16);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.rowoff, B * ws.maxc * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.candlist, B * ws.maxcand * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.ncand, B * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.rowmin, B * P * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.rowmax, B * P * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.cand_rec, B * ws.maxcand * sizeof(vtd_detection));
    rc = rc ? rc : pp_alloc(pp, (void**)&ws.cand_valid, B * ws.maxcand * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&pp->orig_w_dev, B * 4);
    rc = rc ? rc : pp_alloc(pp, (void**)&pp->orig_h_dev, B * 4);
    if (rc) {
        for (int i = 0; i < pp->nblocks; ++i) (void)hipFree(pp->blocks[i]);
        delete pp;
        return rc;
    }
    *out = pp;
    return 0;
}

void vtd_postproc_destroy(vtd_postproc* pp) {
    if (!pp) return;
    for (int i = 0; i < pp->nblocks; ++i) (void)hipFree(pp->blocks[i]);
    delete pp;
}

// Device -> pinned host copy as a kernel (16-byte stores straight into the mapped host buffer).  Why not hipMemcpyAsync: on the
// post-process side stream, behind a cross-stream event wait, the third device-to-host copy after the device had been idle was seen to
// BLOCK the host until the copy could run -- 5.6 ms once per drained pipeline, i.e. once per 20-step bench window (bench.py
// VTD_BENCH_STAMPS=2 + line timers in TextDetector.submit_batch).  A kernel launch never waits for the GPU.  The caller orders the host
// behind the copy with an event on `stream`, as it did for the asynchronous copy.
namespace {
__global__ __launch_bounds__(256) void pp_copy_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                              const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, int tail) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i < tail) dst_tail[i] = src_tail[i];
}
}  // namespace

int vtd_copy_to_pinned_host(const void* src_dev, void* dst_pinned_host, int64_t bytes, vtd_stream stream) {
    if (!src_dev || !dst_pinned_host || bytes <= 0 || (((uintptr_t)src_dev | (uintptr_t)dst_pinned_host) & 15)) return -1100;
    void* mapped = nullptr;
    VTD_HIP_CHECK(hipHostGetDevicePointer(&mapped, dst_pinned_host, 0));   // fails for memory that is not pinned: the caller falls back
    const int64_t n16 = bytes >> 4;
    const int tail = (int)(bytes & 15);
    const int64_t blocks = std::max<int64_t>((std::max<int64_t>(n16, tail) + 255) / 256, 1);
    hipLaunchKernelGGL(pp_copy_to_host_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_dev, (uint4*)mapped, n16,
                       (const uint8_t*)src_dev + (n16 << 4), (uint8_t*)mapped + (n16 << 4), tail);
    return -(int)hipGetLastError();
}

int vtd_postproc_run(vtd_postproc* pp, const float* prob_dev, int n, const int32_t* orig_w_host, const int32_t* orig_h_host,
                     float threshold, vtd_detection* out_dev, int32_t* counts_dev, vtd_stream stream) {
    if (!pp || !prob_dev || !orig_w_host || !orig_h_host || !out_dev || !counts_dev) return -1100;
    if (n <= 0 || n > pp->max_batch) return -1105;
    for (int i = 0; i < n; ++i)
        if (orig_w_host[i] <= 0 || orig_h_host[i] <= 0) return -1100;
    hipStream_t s = (hipStream_t)stream;
    // Frame sizes rarely change between calls: the device copy is refreshed only when they do.  (An asynchronous copy from
    // pageable memory makes the host wait for the stream to reach it, i.e. for the whole detector in front of this call,
    // and that would serialise the caller's batch pipeline.)  On a change: drain the stream, then copy synchronously.
    bool same = (int)pp->orig_shadow.size() >= 2 * n;
    for (int i = 0; same && i < n; ++i) same = pp->orig_shadow[2 * i] == orig_w_host[i] && pp->orig_shadow[2 * i + 1] == orig_h_host[i];
    if (!same) {
        VTD_HIP_CHECK(hipDeviceSynchronize());
        VTD_HIP_CHECK(hipMemcpy(pp->orig_w_dev, orig_w_host, n * 4, hipMemcpyHostToDevice));
        VTD_HIP_CHECK(hipMemcpy(pp->orig_h_dev, orig_h_host, n * 4, hipMemcpyHostToDevice));
        if ((int)pp->orig_shadow.size() < 2 * n) pp->orig_shadow.resize(2 * n, 0);
        for (int i = 0; i < n; ++i) { pp->orig_shadow[2 * i] = orig_w_host[i]; pp->orig_shadow[2 * i + 1] = orig_h_host[i]; }
    }
    PostWs ws = pp->ws;
    ws.prob = prob_dev; ws.n = n; ws.thr = threshold;
    ws.orig_w = pp->orig_w_dev; ws.orig_h = pp->orig_h_dev;
    ws.out = out_dev; ws.out_count = counts_dev;
    const int64_t words = (int64_t)n * ws.h * ws.wpr;
    const int iblocks = (int)std::min<int64_t>(((int64_t)n * ws.h + 3) / 4, 256 * 16);  // one wave per row
    const int wblocks = (int)std::min<int64_t>((words + 255) / 256, 256 * 16);      // one thread per word
    // VTD_PP_FUSED=1: one workgroup per frame for everything up to the row tables instead of the ten launches (bit-identical results:
    // tests/test_gpu_postprocess.py runs both).  Measured on one box, three alternating repetitions of the default bench line
    // (tools/gpu_pp.sh): 12.90-12.97 k frames/s sustained against 13.02-13.08 k with the ten launches, the head entry's in-situ time equal
    // within the spread (465-476 against 472-483 us) -- the frame kernel holds its CUs for 650 us per batch where the chain takes 300, and
    // the thin launches were not what stretches the wide kernels.  The ten launches stay the default.
    const char* fe = getenv("VTD_PP_FUSED");
    if (fe && fe[0] == '1') {
        hipLaunchKernelGGL(pp_frame_kernel, dim3(n), dim3(SCAN_THREADS), 0, s, ws);
    } else {
        hipLaunchKernelGGL(pp_init, dim3(iblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_merge_fg_bg, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_classify, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_merge_inside, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_count_roots, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_scan_slices, dim3(n), dim3(SCAN_THREADS), 0, s, ws);
        hipLaunchKernelGGL(pp_number_components, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_stats, dim3(wblocks), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(pp_candidates, dim3(n), dim3(SCAN_THREADS), 0, s, ws);
        hipLaunchKernelGGL(pp_row_extents, dim3(wblocks), dim3(256), 0, s, ws);
    }
    const size_t box_lds = ((size_t)3 * ws.h + (size_t)5 * (2 * ws.h + 2)) * sizeof(int);  // row table, flags, hull points, vect, inv_len
    if (box_lds > 150 * 1024) return -1011;
    static bool box_attr = false;
    if (!box_attr) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)pp_boxes, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        box_attr = true;
    }
    hipLaunchKernelGGL(pp_boxes, dim3(std::min(ws.maxcand, 128), n), dim3(BOX_THREADS), box_lds, s, ws);
    hipLaunchKernelGGL(pp_emit, dim3(n), dim3(SCAN_THREADS), 0, s, ws);
    return -(int)hipGetLastError();
}

}  // extern "C"
