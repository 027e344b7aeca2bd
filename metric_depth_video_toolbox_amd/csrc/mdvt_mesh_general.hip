// mdvt_mesh_general.hip -- MESH MODE, general path (pose / convergence / K != Krender, and pure-shift frames too wide
// for the LDS row kernels): the rasteriser stage between k_mesh_vertices_general and k_resolve_general.
//
//   replaces dmt.render (dmt:1422-1572) for the triangles of dmt:1243-1254 after the transforms of sr:615-619,
//   724-725, 832-836; z-buffer words and tie rule: see mdvt_device.h.
//
// The grid mesh is ~4 million triangles per 1080p eye, of which all but ~0.5 % cover a handful of pixel centres: the
// rest are the rubber sheet across depth edges.  The kernel that owns a cell therefore only knows ONE kind of
// triangle: snapped extent below 2^13 sub-pixels (32 px) and at most kSmallBox pixel centres in its box.  For that
// kind every edge value fits 32 bits and every product is a 24-bit multiply (mdvt_device.h, TriSmall) -- the
// 32x32 -> 64 multiplies of the generic set-up are quarter rate and were two thirds of this kernel's instructions.
// Everything else goes to a queue as 8 bytes (draw id, frame slot / eye); k_mesh_raster_queue reads the three vertex
// records of a queued triangle back and rasterises it generically, 16 lanes per triangle.  The queue is one segment
// per (frame slot, cell row) with its own counter -- a single counter for the launch serialised ~10^5 returning
// atomics per frame on one address (150 us per 1080p frame when the rubber sheet is not filtered out) -- and every
// segment holds all four triangles of every cell of its row, so nothing can overflow.
#include "mdvt_device.h"
#include <stdio.h>
#include <vector>

namespace mdvt {
namespace MDVT_GRID {      // one copy per sub-pixel grid (mdvt_internal.h)

namespace {

constexpr int kHugeArea = 8192;    // pixel centres in the box of a queued triangle above which its rows are dealt over the chip
constexpr int kHugeRows = 16;      // rows per entry of the huge list
constexpr int kSmallBox = 12;      // pixel centres in the bounding box of a triangle the owning lane walks itself

// Where the fragments of one (frame slot, eye) go.
struct FragOut {
    u64* keys; u64* cbuf;
    uint32_t* tiles; uint32_t* flag;
    uint32_t parity;
    int tiles_x;
};
__device__ __forceinline__ FragOut frag_out(const RenderArgs& a, int slot, int eye)
{
    FragOut f;
    f.keys = a.keys[eye] + (size_t)slot * a.ws_stride_px;
    f.cbuf = a.cbuf[eye] + (size_t)slot * a.ws_stride_px;
    f.tiles = a.tie_tiles + ((size_t)slot * 2 + (size_t)eye) * (size_t)a.tie_words;
    f.flag = a.tie_flag + slot;
    f.parity = (a.key_parity >> slot) & 1u;
    f.tiles_x = a.tie_tiles_x;
    return f;
}

// What a first-pass fragment's post returned is looked at just before the lane's next post (or at the kernel's end).  The post is a
// RETURNING atomic -- the word it replaced is what tells a tie -- a round trip to the L2 that costs the rasterisers ~12 us per 1080p
// frame when it is waited for at once.  Settling the pending word BEFORE the next post frees its registers for the next return (the
// atomic writes them directly; with the opposite order the compiler copies the pair and waits for it right behind the atomic), so
// the round trip overlaps the next fragment's set-up and shading: half of the cost back (profiles/r04_colour_keys.md; batches of
// 2-4 posts per lane and an XCD-aware block order did not help).
struct Pending {
    u64 old, mine;
    uint32_t o, se;      // pixel index (~0u: nothing pending), frame slot << 1 | eye
};
__device__ __forceinline__ Pending pending_none() { Pending p; p.old = p.mine = 0ull; p.o = ~0u; p.se = 0u; return p; }
__device__ __forceinline__ void pending_settle(const RenderArgs& a, const Pending& p)
{
    if (p.o == ~0u) return;
    const int slot = (int)(p.se >> 1);
    const uint32_t parity = (a.key_parity >> slot) & 1u;
    if (zkey_colour_conflict(p.old, p.mine, parity) || ((MDVT_DEBUG_SKIP(a) & 32) && zkey_covered(p.old, parity))) {   // (bit 5: test hook,
        const FragOut f = frag_out(a, slot, (int)(p.se & 1u));                                                         //  tuning build)
        zkey_mark_tied(&f.keys[p.o], parity);
        f.cbuf[p.o] = ~0ull;                                // (every marker stores the same value; the second pass is a later kernel)
        *f.flag = 1u;
        const int px = (int)(p.o % (uint32_t)a.W), py = (int)(p.o / (uint32_t)a.W);
        const int t = (py / kTieTile) * f.tiles_x + px / kTieTile;
        atomicOr(&f.tiles[t >> 5], 1u << (t & 31));
    }
}

// One shaded fragment.  MODE 0 (first pass): nearest fragment's depth and colour into the pixel's key word; a fragment that finds
// its own depth there with another colour marks the pixel (mdvt_device.h, colour keys), its tile and its frame -- when it is
// settled (pending_settle: by the next fragment of the lane, or by the kernel's end).  MODE 1 (second pass, marked frames only):
// at a marked pixel, the fragments at the word's depth compete by draw id in the side word.
template <int MODE>
__device__ __forceinline__ void mesh_global_fragment(const RenderArgs& a, int slot, int eye, const FragOut& f, Pending& pd, int W, int px, int py,
                                                     float q0, float q1, float q2, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t did)
{
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));     // (the colour conversions belong to the fragment, not to every triangle's set-up)
    const uint32_t o = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    const float iz = (q0 + q1) + q2;
    const uint32_t rgb = shade_px(q0, q1, q2, rcp_exact(iz), c0, c1, c2);
    const u64 mine = zkey_word<true>(f.parity, __float_as_uint(iz), kNoTie | rgb);      // iz > 0: its bits are a 31-bit order key
    if (MODE == 0) {
#ifdef MDVT_ABLATE_POST           // (a build of its own, timing ablation: the fragment without its post -- as a run-time hook it cost the walks registers)
        if (MDVT_ABLATE_POST) { asm volatile("" :: "v"(mine)); return; }
#endif
        pending_settle(a, pd);            // (before the post: the pending word's registers are then free to take the next return)
        pd.old = zkey_post_word(&f.keys[o], f.parity, mine);
        pd.mine = mine; pd.o = o; pd.se = (uint32_t)slot * 2u + (uint32_t)eye;
    } else {
        const u64 key = f.keys[o];
        if (zkey_covered(key, f.parity) && zkey_is_tied(key, f.parity) && ((key ^ mine) >> 32) == 0ull)
            atomicMin(&f.cbuf[o], ((u64)did << 32) | rgb);
    }
}

// ---- The projected vertex records {X, Y (snapped), 1/Z', rgb}: made where they are used (r05) ----------------------------------
// Until r04 a vertex pass (k_mesh_vertices_general) projected every vertex once per eye into a 16-byte record -- 66 MB of stores
// per 1080p frame, read back 2 to 5 times by the rasterisers (161 MB per frame of the product default, 596 MB per 4K frame
// under a pose), 32 B/px of workspace per slot and the second longest kernel of every general mesh path (29 us per 1080p frame)
// for ~150 instructions per vertex.  Now a workgroup of the cell walks computes the two vertex rows of its cells from the source
// frame (6 B per vertex, read through the L2 by the row above and the row below) into LDS -- 64 threads, 64 columns, 63 cells, so no
// second pass over a 65th column -- and the queue / huge walks recompute the three vertices of their triangle.  The same device
// functions in the same order as the vertex pass ran them: the same bits.
// (CONV: every frame of the launch is convergence-only -- k_mesh_raster_conv's launches: vertex_conv_only)
template <bool CONV>
__device__ __forceinline__ void vertex_records(const RenderArgs& a, const FrameDev& fp, int f, int i, int j, uint4& r0, uint4& r1)
{
    const uint32_t dpx = load_px_bytes(a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch, j);
    const uint32_t rgb = load_px_bytes(a.color + (size_t)f * a.color_stride + (size_t)i * a.color_pitch, j);
    const float z = decode_z(code16_of(dpx), fp.mult, fp.scale);
    const float gx = (float)j * fp.sx, gy = (float)i * fp.sy;
    float xc, yc;
    camera_point(fp, gx, gy, z, xc, yc);
    const Vert v0 = CONV ? vertex_conv_only(fp, fp.M[0], xc, yc, z) : vertex_for_eye(fp, 0, gx, gy, z, xc, yc);
    const Vert v1 = CONV ? vertex_conv_only(fp, fp.M[1], xc, yc, z) : vertex_for_eye(fp, 1, gx, gy, z, xc, yc);
    const float iz0 = v0.ok ? rcp_exact(v0.z) : 0.0f, iz1 = v1.ok ? rcp_exact(v1.z) : 0.0f;      // 0 flags a vertex behind the near plane
    r0 = make_uint4((uint32_t)snap(v0.u), (uint32_t)snap(v0.v), __float_as_uint(iz0), rgb);
    r1 = make_uint4((uint32_t)snap(v1.u), (uint32_t)snap(v1.v), __float_as_uint(iz1), rgb);
}
__device__ __forceinline__ uint4 vertex_record_eye(const RenderArgs& a, const FrameDev& fp, int f, int i, int j, int eye)
{
    const uint32_t dpx = load_px_bytes(a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch, j);
    const uint32_t rgb = load_px_bytes(a.color + (size_t)f * a.color_stride + (size_t)i * a.color_pitch, j);
    const float z = decode_z(code16_of(dpx), fp.mult, fp.scale);
    const float gx = (float)j * fp.sx, gy = (float)i * fp.sy;
    float xc, yc;
    camera_point(fp, gx, gy, z, xc, yc);
    const Vert v = vertex_for_eye(fp, eye, gx, gy, z, xc, yc);
    const float iz = v.ok ? rcp_exact(v.z) : 0.0f;
    return make_uint4((uint32_t)snap(v.u), (uint32_t)snap(v.v), __float_as_uint(iz), rgb);
}

__device__ __forceinline__ uint4 shfl_u4(const uint4& v, int src)
{
    return make_uint4((uint32_t)__shfl((int)v.x, src), (uint32_t)__shfl((int)v.y, src), (uint32_t)__shfl((int)v.z, src), (uint32_t)__shfl((int)v.w, src));
}
template <typename T> __device__ __forceinline__ uint4 readlane_u4(const uint4& v, T lane)
{
    return make_uint4((uint32_t)__builtin_amdgcn_readlane((int)v.x, lane), (uint32_t)__builtin_amdgcn_readlane((int)v.y, lane),
                      (uint32_t)__builtin_amdgcn_readlane((int)v.z, lane), (uint32_t)__builtin_amdgcn_readlane((int)v.w, lane));
}

constexpr int kCellTPB = 64;          // threads per workgroup of the cell walks = vertex columns it stages
constexpr int kCellsWG = kCellTPB - 1;   // cells per workgroup: between its 64 columns
// Vertex row `row` of columns j0 .. j0 + 63, both eyes, into ring slot `slot`: sv[eye][slot][column].
template <bool CONV>
__device__ __forceinline__ void stage_vertex_row(const RenderArgs& a, int fr, int row, int j0, uint4 (&sv)[2][2][kCellTPB], int slot)
{
    const int t = threadIdx.x, f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const int jc = min(j0 + t, a.W - 1);                                   // (clamped: columns past the row end are never used)
    uint4 r0, r1;
    vertex_records<CONV>(a, fp, f, row, jc, r0, r1);
    sv[0][slot][t] = r0; sv[1][slot][t] = r1;
}
// A workgroup of the cell walks takes kRowsWG rows of cells of its 63 columns, top to bottom, its vertex rows in a ring of two: the
// row between two rows of cells is worked out once, not by the workgroup above it AND the one below (r05, second step: with the
// vertex programme inside the walk every vertex was computed twice per eye, ~300 of the ~1 400 instructions a cell costs).
#ifndef MDVT_ROWS_WG
#define MDVT_ROWS_WG 4
#endif
#ifndef MDVT_ROWS_WG_CONV_EDGES
#define MDVT_ROWS_WG_CONV_EDGES MDVT_ROWS_WG
#endif
__host__ __device__ constexpr int rows_wg(int flags, bool conv) { return (conv && (flags & 2)) ? MDVT_ROWS_WG_CONV_EDGES : MDVT_ROWS_WG; }
// ... and the occupancy each is compiled for (waves per SIMD; 1 = whatever the registers come to).  What the row loop costs is
// registers: the per-lane addresses its body leads to (source pixels, filter flags, key planes) are hoisted out of it and live across
// the whole walk -- 80 VGPRs (six waves per SIMD) became 87-101 (five, four), which ate the gain and more for the scanline walk with
// edge removal.  With the block's column and frame made opaque per row (an empty asm: the addresses are worked out again, a few
// integer operations) the scanline walks fit 80 (20 bytes of scratch with edge removal) and the triangle walks take 80 / 83.  Measured, same box each:
// mesh + convergence 14.8 (one row) -> 16.3 (four rows) -> 18.0 k frames/s (80 VGPRs, convergence-only vertex programme); product
// default (scanline walk with edge removal) 14.27 k at one row -> 14.70 k at four rows and 80 VGPRs (two rows: 14.43 k); mesh under a
// pose 12.7 -> 13.8 k; C4 mesh 2.29 -> 2.42 k; eight rows give no more.
#ifndef MDVT_SMALL2_WAVES
#define MDVT_SMALL2_WAVES 1
#endif
__host__ __device__ constexpr int cell_waves(int flags, bool conv) { return conv ? 6 : ((flags & 2) ? MDVT_SMALL2_WAVES : 6); }
// The block of cells of a workgroup of the one-thread-per-cell kernels: row-major over the frame's cells, grid x = cell_blocks.
// (Measured and not kept, r04: an XCD-aware deal -- workgroup b runs on XCD b % 8, each XCD with its own L2, so each XCD took a
// contiguous eighth of the blocks and the second reader of a vertex-record row found it in the L2 the first one had filled.  The
// rasteriser's fetch traffic fell from 145 to 71 MB per 1080p frame and its time did not move, 6 % slower under a pose: these
// kernels wait for their atomics, not for bytes.)
__host__ __device__ __forceinline__ uint32_t cell_blocks(int W, int H, int rows) { return (uint32_t)((W - 1 + kCellsWG - 1) / kCellsWG) * (uint32_t)((H - 1 + rows - 1) / rows); }
// block v of a frame -> its column block bx and its first row of cells i0
__device__ __forceinline__ void cell_block_of(int W, uint32_t v, int rows, int& bx, int& i0)
{
    const uint32_t nbx = (uint32_t)((W - 1 + kCellsWG - 1) / kCellsWG);
    bx = (int)(v % nbx); i0 = (int)(v / nbx) * rows;
}

// The queue's two levels of counters.  A wave that appends n triangles to a segment (one returning atomic on the segment's counter:
// its entries' places) also adds n to the counter of the block of 2^bigq_shift segments the segment belongs to -- a posted atomic,
// nothing waits for it.  The queue walk sums the few thousand block counters up in LDS and finds the segment of a global entry from
// there (queue_locate).  Until r05 a kernel of ONE workgroup between the cell walk and the queue walk turned the segment counters
// into prefix sums (a dependent launch, 5 us per 1080p frame of the product default in the profile), and every entry of the queue
// walk began with a 15-step binary search through them in the L2.
__device__ __forceinline__ void queue_count_coarse(const RenderArgs& a, size_t seg, uint32_t n)
{
    atomicAdd(&a.bigq_coarse[seg >> a.bigq_shift], n);
}
constexpr int kQueueCoarseMax = 4096;          // block counters per launch set at most (launch_mesh_raster_general picks bigq_shift)
inline int queue_shift_for(int nseg) { int sh = 4; while (((nseg + (1 << sh) - 1) >> sh) > kQueueCoarseMax) ++sh; return sh; }

// cpre[0 .. nc]: exclusive prefix sums of the block counters, by the whole workgroup (any size); cpre[nc] = queued triangles.
__device__ __forceinline__ void queue_prefix_lds(const RenderArgs& a, int nseg, uint32_t* cpre)
{
    const int nc = (nseg + (1 << a.bigq_shift) - 1) >> a.bigq_shift;
    const int t = (int)threadIdx.x, nt = (int)blockDim.x;
    const int per = (nc + nt - 1) / nt, lo = min(t * per, nc), hi = min(lo + per, nc);
    uint32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += a.bigq_coarse[k];
    // (cpre[nc + 1 ..] is scratch for the per-thread sums: the launcher sized it nc + 1 + threads)
    uint32_t* part = cpre + nc + 1;
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {
        const uint32_t u = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int k = lo; k < hi; ++k) { cpre[k] = run; run += a.bigq_coarse[k]; }
    if (t == nt - 1) cpre[nc] = part[nt - 1];
    __syncthreads();
}

// Segment and place within it of global queue entry g (< cpre[nc]); the 16 lanes of a group call it together with the same g.
__device__ __forceinline__ void queue_locate(const RenderArgs& a, int nseg, const uint32_t* cpre, uint32_t g, int sub, int& seg_out, uint32_t& k_out)
{
    const int nc = (nseg + (1 << a.bigq_shift) - 1) >> a.bigq_shift;
    int lo_b = 0, hi_b = nc - 1;                                   // last block with cpre[b] <= g
    while (lo_b < hi_b) {
        const int mid = (lo_b + hi_b + 1) >> 1;
        if (cpre[mid] <= g) lo_b = mid; else hi_b = mid - 1;
    }
    uint32_t rem = g - cpre[lo_b];
    const int s0 = lo_b << a.bigq_shift, s1 = min(s0 + (1 << a.bigq_shift), nseg);
    const int row15 = (int)(threadIdx.x & 63u) | 15;
    seg_out = s1 - 1; k_out = 0u;
    for (int c = s0; c < s1; c += 16) {                            // the block's segment counters, 16 at a time: one per lane
        const int sgm = c + sub;
        const uint32_t cnt = sgm < s1 ? a.bigq_count[sgm] : 0u;
        uint32_t inc = cnt;                                        // inclusive sums along the group's 16 lanes (one DPP row)
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
        const uint32_t tot = (uint32_t)__shfl((int)inc, row15);
        if (rem < tot) {
            const uint32_t m = (uint32_t)(__ballot(inc > rem) >> ((threadIdx.x & 63u) & ~15u)) & 0xFFFFu;     // this group's lanes
            const int idx = __ffs((int)m) - 1;
            const int src = (int)((threadIdx.x & 63u) & ~15u) + idx;
            seg_out = c + idx;
            k_out = rem - ((uint32_t)__shfl((int)inc, src) - (uint32_t)__shfl((int)cnt, src));
            return;
        }
        rem -= tot;
    }
}

// second pass: does the pixel box touch a tile with a marked pixel?
__device__ __forceinline__ bool tie_tiles_hit(const FragOut& f, int bx0, int by0, int bx1, int by1)
{
    bool hit = false;
    for (int ty = by0 / kTieTile; ty <= by1 / kTieTile; ++ty)
        for (int tx = bx0 / kTieTile; tx <= bx1 / kTieTile; ++tx) {
            const int t = ty * f.tiles_x + tx;
            hit |= ((f.tiles[t >> 5] >> (t & 31)) & 1u) != 0u;
        }
    return hit;
}

}  // namespace

// One thread per cell, both triangles, both eyes.  MODE as mesh_global_fragment's; the second pass runs for the frames with a marked
// pixel only, and only the triangles whose pixel box touches a marked tile get as far as their set-up.
// (the caller has staged the row's two vertex rows: sv[eye][top] above, sv[eye][top ^ 1] below, and synchronised)
template <int FLAGS, int MODE>
__device__ __forceinline__ void mesh_raster_small_block(const RenderArgs& a, int fr, int bx, int i, uint4 (&sv)[2][2][kCellTPB], int top, Pending (&pds)[2])
{
    constexpr bool EDGES = FLAGS & 2;
    const int W = a.W, H = a.H;
    const int j = bx * kCellsWG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool act = j < W - 1 && threadIdx.x < kCellsWG;
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    // the 2 x kCellTPB vertex records of both eyes, computed once per workgroup
    uint32_t inv0 = 0, inv1 = 0;
    if (EDGES && act) {
        const uint8_t* tinv = a.tri_invalid + (size_t)fr * a.ws_stride_tri + (size_t)i * (W - 1) + j;
        inv0 = tinv[0]; inv1 = tinv[ncell];
    }
    bool tq[2][2] = {{false, false}, {false, false}};          // [eye][triangle of the cell]: goes to the queue
    // (unrolled: each eye has its own pending word, so the left eye's last post is in flight while the right eye is rasterised)
#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
        Pending& pd = pds[eye];
        const FragOut fo = frag_out(a, fr, eye);
        uint4 A = make_uint4(0, 0, 0, 0), B = A, Cv = A, D = A;
        if (act) {
            const int t = threadIdx.x;
            A = sv[eye][top][t]; D = sv[eye][top][t + 1]; B = sv[eye][top ^ 1][t]; Cv = sv[eye][top ^ 1][t + 1];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            // tri1 = (v[i,j], v[i+1,j], v[i+1,j+1]); tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])   (dmt:1243-1254)
            const uint4 v1 = pass == 0 ? B : Cv, v2 = pass == 0 ? Cv : D;
            const uint32_t did = draw_id_global(pass, i, j);
            bool toq = false;
            if (act && !(pass == 0 ? inv0 : inv1)) {
                const int X0 = (int)A.x, Y0 = (int)A.y, X1 = (int)v1.x, Y1 = (int)v1.y, X2 = (int)v2.x, Y2 = (int)v2.y;
                const float iz0 = __uint_as_float(A.z), iz1 = __uint_as_float(v1.z), iz2 = __uint_as_float(v2.z);
                if (iz0 > 0.0f && iz1 > 0.0f && iz2 > 0.0f) {          // else: near plane, whole triangle dropped
                    const int mnX = min3i(X0, X1, X2), mxX = max3i(X0, X1, X2), mnY = min3i(Y0, Y1, Y2), mxY = max3i(Y0, Y1, Y2);
                    int bx0 = floordiv_subpix(mnX - kSubpix / 2 + kSubpix - 1), bx1 = floordiv_subpix(mxX - kSubpix / 2);
                    int by0 = floordiv_subpix(mnY - kSubpix / 2 + kSubpix - 1), by1 = floordiv_subpix(mxY - kSubpix / 2);
                    bx0 = max(bx0, 0); by0 = max(by0, 0); bx1 = min(bx1, W - 1); by1 = min(by1, H - 1);
                    if (bx1 >= bx0 && by1 >= by0) {                      // else: no pixel centre in the box
                        // (a small extent bounds the box at 33 x 33: the count fits any integer)
                        if (max(mxX - mnX, mxY - mnY) >= kSmallTriExtent || (bx1 - bx0 + 1) * (by1 - by0 + 1) > kSmallBox) {
                            toq = MODE == 0;             // (second pass: nothing is queued -- the first pass's lists are walked again)
                        } else if (MODE == 0 || tie_tiles_hit(fo, bx0, by0, bx1, by1)) {
                            TriSmall ts;
                            if (tri_small_setup(ts, X0, Y0, iz0, X1, Y1, iz1, X2, Y2, iz2, a.cull) && !(MDVT_DEBUG_SKIP(a) & 16)) {
                                TriWalk32 row = tri_small_start(ts, bx0, by0);
                                for (int py = by0; py <= by1; ++py) {
                                    TriWalk32 w = row;
                                    for (int px = bx0; px <= bx1; ++px) {
                                        if (tri_small_inside(ts, w)) {
                                            float q0, q1, q2;
                                            tri_small_weights(ts, w, q0, q1, q2);
                                            mesh_global_fragment<MODE>(a, fr, eye, fo, pd, W, px, py, q0, q1, q2, A.w, v1.w, v2.w, did);
                                        }
                                        tri_small_right(ts, w);
                                    }
                                    tri_small_down(ts, row);
                                }
                            }
                        }
                    }
                }
            }
            tq[eye][pass] = toq;
        }
    }
    // The workgroup's (= wave's) queued triangles of both eyes in ONE append: one returning atomic on the segment's counter and one
    // posted atomic on its block's (r05; an append per (eye, triangle of the cell) was four of each, and with everything in flight
    // working on the same few rows those counters are where the rasteriser's atomics queue up: +10 % on its time under a pose).
    if (MODE == 0) {
        u64 m[4];
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = __ballot(tq[k >> 1][k & 1]); total += (uint32_t)__popcll(m[k]); }
        if (total) {                                                      // (wave uniform)
            const size_t seg = (size_t)fr * H + i;                        // this row's segment: 4 (W - 1) entries at most
            uint32_t base = 0;
            if (lane == 0) { queue_count_coarse(a, seg, total); base = atomicAdd(&a.bigq_count[seg], total); }
            base = __shfl(base, 0);
            uint2* q = (uint2*)a.bigq + seg * (size_t)(4 * W);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tq[k >> 1][k & 1])
                    q[base + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull))] = make_uint2(draw_id_global(k & 1, i, j), (uint32_t)fr * 2u + (uint32_t)(k >> 1));
                base += (uint32_t)__popcll(m[k]);
            }
        }
    }
}

// First pass: a workgroup per block of 63 x kRowsWG cells (grid: cell_blocks x 1 x frames).
template <int FLAGS>
__global__ void __launch_bounds__(kCellTPB, cell_waves(FLAGS, false)) k_mesh_raster_small(RenderArgs a)
{
    __shared__ uint4 sv[2][2][kCellTPB];
    Pending pds[2] = {pending_none(), pending_none()};
    constexpr int kRowsWG = rows_wg(FLAGS, false);
    if (blockIdx.x >= cell_blocks(a.W, a.H, kRowsWG)) return;
    int bx, i0;
    cell_block_of(a.W, blockIdx.x, kRowsWG, bx, i0);
    const int fr = (int)blockIdx.z;
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (r05 diagnosis, tuning build)
    stage_vertex_row<false>(a, fr, i0, bx * kCellsWG, sv, 0);
#pragma unroll 1
    for (int r = 0; r < kRowsWG && i0 + r < a.H - 1; ++r) {
        int bxr = bx, frr = fr;                            // (opaque per row: see k_mesh_raster_conv)
        asm volatile("" : "+s"(bxr), "+s"(frr));
        stage_vertex_row<false>(a, frr, i0 + r + 1, bxr * kCellsWG, sv, (r + 1) & 1);
        __syncthreads();
        mesh_raster_small_block<FLAGS, 0>(a, frr, bxr, i0 + r, sv, r & 1, pds);
        __syncthreads();                                   // (the row above is replaced next: its last readers are through)
    }
    pending_settle(a, pds[0]);
    pending_settle(a, pds[1]);
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// ---- The same stage for frames whose vertex rows stay (almost) horizontal on screen: convergence only ---------------------
// With nothing but a toe-in (FrameDev.conv_band: sr:707-726, no pose, K == Krender) the projected row of a vertex does not
// depend on its depth: a cell is still a slightly sheared box, at most a sub-pixel taller on one side than on the other, and
// nearly every scanline that meets it enters through its left column edge and leaves through its right one, crossing the
// diagonal in between -- never its top or bottom edge.  On such a scanline the covered pixels are
//     [P(left crossing), P(right crossing))  split at P(diagonal crossing),   P(k, h) = ceil((k - 128 h) / (256 h))
// (the fill rule of the generic edge functions written out for edges that start above the scanline and end at or below it; h =
// the edge's own height), and a fragment's three edge functions are 32-bit values from 24-bit multiplies.  That is ~150
// instructions per cell and eye where the triangle-by-triangle set-up and walk of k_mesh_raster_small needs 2 x 211.  Whatever
// does not fit -- a scanline through a cell's top or bottom edge (a few per thousand cells), twisted or folded cells, cells
// behind the near plane or beyond +-2^19 sub-pixels, stretched cells -- is listed per workgroup and goes through the generic
// small-triangle / queue code afterwards, every lane a listed triangle.  Fragments are the same words either way (a fragment
// posted twice changes nothing: the key is a minimum, the side buffer word is identical).
__device__ __forceinline__ int conv_first_pixel(int k, int h, int W)
{
    const int D = h * kSubpix;
    int n = k + h * (kSubpix / 2) - 1;
    n = n < 0 ? 0 : n;
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)D));
    const int rem = n - __mul24(q, D);
    q += rem < 0 ? -1 : (rem >= D ? 1 : 0);
    return q > W ? W : q;
}

constexpr int kConvTPB = 64;          // cells (threads) per workgroup of k_mesh_raster_conv
// MODE as mesh_global_fragment's (second pass: the same classification of the cells as in the first, so that every fragment of the
// first pass is met again -- here, or in the first pass's lists of queued triangles, which this pass does not add to).
template <int FLAGS, int MODE>
__device__ __forceinline__ void mesh_raster_conv_block(const RenderArgs& a, int fr, int bx, int i, uint4 (&sv)[2][2][kConvTPB], int top,
                                                       uint32_t (&glist)[2 * kConvTPB], uint32_t& gcount, Pending& pd)
{
    constexpr bool EDGES = FLAGS & 2;
    constexpr int kCoord = 1 << 19, kMaxH = 512, kMaxSpan = 12;
    const int W = a.W, H = a.H;
    const int j = bx * kCellsWG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool act = j < W - 1 && threadIdx.x < kCellsWG;
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    uint32_t inv0 = 0, inv1 = 0;
    if (EDGES && act) {
        const uint8_t* tinv = a.tri_invalid + (size_t)fr * a.ws_stride_tri + (size_t)i * (W - 1) + j;
        inv0 = tinv[0]; inv1 = tinv[ncell];
    }
    if (threadIdx.x == 0) gcount = 0u;
    __syncthreads();                                       // (also: the caller's vertex rows are staged)
    const uint32_t cull = (uint32_t)a.cull;
    // (not unrolled with a pending word per eye, as k_mesh_raster_small is: measured, product default 2640 -> 2750 us per 32 frames)
#pragma unroll 1
    for (int eye = 0; eye < 2; ++eye) {
        if (!act || (inv0 && inv1)) continue;
        const FragOut fo = frag_out(a, fr, eye);
        const int t = threadIdx.x;
        const uint4 A = sv[eye][top][t], D = sv[eye][top][t + 1], B = sv[eye][top ^ 1][t], Cv = sv[eye][top ^ 1][t + 1];
        const int XA = (int)A.x, YA = (int)A.y, XB = (int)B.x, YB = (int)B.y, XC = (int)Cv.x, YC = (int)Cv.y, XD = (int)D.x, YD = (int)D.y;
        const float izA = __uint_as_float(A.z), izB = __uint_as_float(B.z), izC = __uint_as_float(Cv.z), izD = __uint_as_float(D.z);
        if (!(izA > 0.0f)) continue;                                   // A is a vertex of both triangles: near plane, both dropped
        bool generic = !(izB > 0.0f && izC > 0.0f && izD > 0.0f);
        const int hAB = YB - YA, hAC = YC - YA, hDC = YC - YD;
        const uint32_t inr = (uint32_t)(XA + kCoord) | (uint32_t)(XB + kCoord) | (uint32_t)(XC + kCoord) | (uint32_t)(XD + kCoord);
        generic |= (inr >> 20) != 0u || hAB <= 0 || hAB >= kMaxH || hAC <= 0 || hAC >= kMaxH || hDC <= 0 || hDC >= kMaxH;
        if (!generic) {
            int k_lo = floordiv_subpix(min(YA, YD) - kSubpix / 2 + kSubpix), k_hi = floordiv_subpix(max(YB, YC) - kSubpix / 2);      // centres in (top, bottom]
            k_lo = max(k_lo, 0); k_hi = min(k_hi, H - 1);
            if (k_hi - k_lo > 2) generic = true;
            else {
                for (int k = k_lo; k <= k_hi; ++k) {
                    const int Yc = k * kSubpix + kSubpix / 2;
                    if (!(YA < Yc && YD < Yc && Yc <= YB && Yc <= YC)) {
                        // the scanline passes through the cell's top or bottom edge -- unless it misses the cell altogether
                        if (!((Yc <= YA && Yc <= YD) || (YB < Yc && YC < Yc))) generic = true;
                        continue;
                    }
                    const int tA = Yc - YA, tD = Yc - YD;
                    const int kAB = __mul24(XB - XA, tA) + __mul24(hAB, XA), kAC = __mul24(XC - XA, tA) + __mul24(hAC, XA);
                    const int kDC = __mul24(XC - XD, tD) + __mul24(hDC, XD);
                    const int pAB = conv_first_pixel(kAB, hAB, W), pAC = conv_first_pixel(kAC, hAC, W), pDC = conv_first_pixel(kDC, hDC, W);
                    const bool regular = pAB <= pDC;
                    const bool mono = regular ? (pAB <= pAC && pAC <= pDC) : (pAC <= pAB && pDC <= pAC);
                    const int plo = regular ? pAB : pDC, n = regular ? pDC - pAB : pAB - pDC;
                    if (!mono || n > kMaxSpan) { generic = true; continue; }
                    for (int px = plo; px < plo + n; ++px) {
                        const bool in1 = (px < pAC) == regular;              // tri1 = (A, B, C) on the B side of the diagonal, tri2 = (A, C, D)
                        if (in1 ? inv0 : inv1) continue;                     // removed by the 89-degree filter (dmt:1372)
                        const int X1 = in1 ? XB : XC, Y1 = in1 ? YB : YC, X2 = in1 ? XC : XD, Y2 = in1 ? YC : YD;
                        const int Xc = px * kSubpix + kSubpix / 2;
                        // the raw edge functions of tri_setup_snapped; inside the triangle all three carry the sign of the doubled area
                        const int w0 = __mul24(X2 - X1, Yc - Y1) - __mul24(Y2 - Y1, Xc - X1);
                        const int w1 = __mul24(XA - X2, Yc - Y2) - __mul24(YA - Y2, Xc - X2);
                        const int w2 = __mul24(X1 - XA, Yc - YA) - __mul24(Y1 - YA, Xc - XA);
                        const int a2 = (w0 + w1) + w2;
                        if (a2 == 0) continue;
                        if (cull && (cull == 1u) != (a2 < 0)) continue;
                        const float ra = rcp_exact((float)(a2 < 0 ? -a2 : a2));
                        const float q0 = ((float)(w0 < 0 ? -w0 : w0) * ra) * izA;
                        const float q1 = ((float)(w1 < 0 ? -w1 : w1) * ra) * (in1 ? izB : izC);
                        const float q2 = ((float)(w2 < 0 ? -w2 : w2) * ra) * (in1 ? izC : izD);
                        mesh_global_fragment<MODE>(a, fr, eye, fo, pd, W, px, k, q0, q1, q2, A.w, in1 ? B.w : Cv.w, in1 ? Cv.w : D.w, draw_id_global(in1 ? 0 : 1, i, j));
                    }
                }
            }
        }
        if (generic) glist[atomicAdd(&gcount, 1u)] = (uint32_t)t | ((uint32_t)eye << 8);
    }
    __syncthreads();
    // ---- the listed cells: k_mesh_raster_small's code, a listed TRIANGLE per lane ----
    const uint32_t ng = 2u * gcount;
    uint32_t nqueued = 0;                                                               // (wave uniform) this workgroup's appends
    for (uint32_t base = 0; base < ng; base += (uint32_t)kConvTPB) {                   // (workgroup uniform)
        const uint32_t idx = base + threadIdx.x;
        const bool on = idx < ng;
        const uint32_t ent = glist[on ? idx >> 1 : 0];
        const int t = (int)(ent & 0xFFu), eye = (int)(ent >> 8), pass = (int)(idx & 1u);
        const int cj = bx * kCellsWG + t;
        const FragOut fo = frag_out(a, fr, eye);
        const uint4 A = sv[eye][top][t], D = sv[eye][top][t + 1], B = sv[eye][top ^ 1][t], Cv = sv[eye][top ^ 1][t + 1];
        const uint4 v1 = pass == 0 ? B : Cv, v2 = pass == 0 ? Cv : D;
        const uint32_t did = draw_id_global(pass, i, cj);
        bool removed = false;
        if (EDGES && on) removed = a.tri_invalid[(size_t)fr * a.ws_stride_tri + (size_t)i * (W - 1) + cj + (pass ? ncell : 0)] != 0;
        bool toq = false;
        if (on && !removed) {
            const int X0 = (int)A.x, Y0 = (int)A.y, X1 = (int)v1.x, Y1 = (int)v1.y, X2 = (int)v2.x, Y2 = (int)v2.y;
            const float iz0 = __uint_as_float(A.z), iz1 = __uint_as_float(v1.z), iz2 = __uint_as_float(v2.z);
            if (iz0 > 0.0f && iz1 > 0.0f && iz2 > 0.0f) {
                const int mnX = min3i(X0, X1, X2), mxX = max3i(X0, X1, X2), mnY = min3i(Y0, Y1, Y2), mxY = max3i(Y0, Y1, Y2);
                int bx0 = floordiv_subpix(mnX - kSubpix / 2 + kSubpix - 1), bx1 = floordiv_subpix(mxX - kSubpix / 2);
                int by0 = floordiv_subpix(mnY - kSubpix / 2 + kSubpix - 1), by1 = floordiv_subpix(mxY - kSubpix / 2);
                bx0 = max(bx0, 0); by0 = max(by0, 0); bx1 = min(bx1, W - 1); by1 = min(by1, H - 1);
                if (bx1 >= bx0 && by1 >= by0) {
                    if (max(mxX - mnX, mxY - mnY) >= kSmallTriExtent || (bx1 - bx0 + 1) * (by1 - by0 + 1) > kSmallBox) {
                        toq = MODE == 0;
                    } else {
                        TriSmall ts;
                        if (tri_small_setup(ts, X0, Y0, iz0, X1, Y1, iz1, X2, Y2, iz2, a.cull)) {
                            TriWalk32 row = tri_small_start(ts, bx0, by0);
                            for (int py = by0; py <= by1; ++py) {
                                TriWalk32 w = row;
                                for (int px = bx0; px <= bx1; ++px) {
                                    if (tri_small_inside(ts, w)) {
                                        float q0, q1, q2;
                                        tri_small_weights(ts, w, q0, q1, q2);
                                        mesh_global_fragment<MODE>(a, fr, eye, fo, pd, W, px, py, q0, q1, q2, A.w, v1.w, v2.w, did);
                                    }
                                    tri_small_right(ts, w);
                                }
                                tri_small_down(ts, row);
                            }
                        }
                    }
                }
            }
        }
        const u64 mq = __ballot(toq);
        if (mq) {
            const size_t seg = (size_t)fr * H + i;
            uint32_t qb = 0;
            const int first = __ffsll((long long)mq) - 1;
            if (lane == first) qb = atomicAdd(&a.bigq_count[seg], (uint32_t)__popcll(mq));
            qb = __shfl(qb, first);
            nqueued += (uint32_t)__popcll(mq);
            if (toq) {
                uint2* q = (uint2*)a.bigq + seg * (size_t)(4 * W);
                q[qb + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull))] = make_uint2(did, (uint32_t)fr * 2u + (uint32_t)eye);
            }
        }
    }
    if (MODE == 0 && nqueued && lane == 0) queue_count_coarse(a, (size_t)fr * H + i, nqueued);      // (one posted atomic per workgroup)
}

template <int FLAGS>
__global__ void __launch_bounds__(kConvTPB, cell_waves(FLAGS, true)) k_mesh_raster_conv(RenderArgs a)
{
    __shared__ uint4 sv[2][2][kConvTPB];
    __shared__ uint32_t glist[2 * kConvTPB];  // cells (thread | eye << 8) for the generic code
    __shared__ uint32_t gcount;
    constexpr int kRowsWG = rows_wg(FLAGS, true);
    if (blockIdx.x >= cell_blocks(a.W, a.H, kRowsWG)) return;
    int bx, i0;
    cell_block_of(a.W, blockIdx.x, kRowsWG, bx, i0);
    const int fr = (int)blockIdx.z;
    Pending pd = pending_none();
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (r05 diagnosis, tuning build)
    stage_vertex_row<true>(a, fr, i0, bx * kCellsWG, sv, 0);
#pragma unroll 1
    for (int r = 0; r < kRowsWG && i0 + r < a.H - 1; ++r) {
        // (the block's column and frame made opaque per row: hoisted out of the loop, the per-lane addresses they lead to -- source
        //  pixels, filter flags, key planes -- live in ~15 VGPRs across the whole walk and cost the kernel a wave per SIMD)
        int bxr = bx, frr = fr;
        asm volatile("" : "+s"(bxr), "+s"(frr));
        stage_vertex_row<true>(a, frr, i0 + r + 1, bxr * kCellsWG, sv, (r + 1) & 1);
        mesh_raster_conv_block<FLAGS, 0>(a, frr, bxr, i0 + r, sv, r & 1, glist, gcount, pd);      // (synchronises before it reads the rows)
        __syncthreads();                                   // (the row above and the list are replaced next)
    }
    pending_settle(a, pd);
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// The queued triangles, dealt over the whole chip (a horizontal depth edge under vertical parallax turns an entire row
// of cells into large triangles: one workgroup per segment would leave that segment's workgroup running alone): 16 lanes
// per triangle, generic 64-bit set-up, rows walked by their own column range.
// (cpre: queue_prefix_lds's sums, in the workgroup's LDS)
template <int MODE>
__device__ __forceinline__ void mesh_queue_walk(const RenderArgs& a, int nseg, const uint32_t* cpre)
{
    const int W = a.W, H = a.H;
    const uint32_t total = cpre[(nseg + (1 << a.bigq_shift) - 1) >> a.bigq_shift];
    const int sub = threadIdx.x & 15;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = (gridDim.x * blockDim.x) >> 4;
    Pending pd = pending_none();
    for (uint32_t g = group; g < total; g += ngroups) {
        int lo_s;                                                   // the segment holding global entry g, and g's place in it
        uint32_t k;
        queue_locate(a, nseg, cpre, g, sub, lo_s, k);
        const uint2* q = (const uint2*)a.bigq + (size_t)lo_s * (size_t)(4 * W);

        uint2 e = q[k];
        if (MODE == 1 && (e.y >> 31)) continue;                        // its row blocks are in the huge list (first pass, below)
        const uint32_t id = e.x;
        const int eye = (int)(e.y & 1u), slot = (int)(e.y >> 1);
        if (MODE == 1 && a.tie_flag[slot] == 0u) continue;              // second pass: the frames with a marked pixel only
        const int pass = (int)(id >> 31), ci = (int)((id >> 16) & 0x7FFFu), cj = (int)(id & 0xFFFFu);
        const int vf = a.frame0 + slot;
        const FrameDev& vfp = a.fp[vf];
        // the triangle's three vertices: one each by the first three lanes of the group (the vertex programme is ~150 instructions a
        // vertex whether one lane runs it or sixteen), handed round the group
        const int vsel = sub < 3 ? sub : 0;
        const uint4 rec = vertex_record_eye(a, vfp, vf, ci + (vsel == 1 || (vsel == 2 && pass == 0) ? 1 : 0),
                                            cj + (vsel == 2 || (vsel == 1 && pass != 0) ? 1 : 0), eye);
        const int gl0 = (int)(threadIdx.x & 63u) & ~15;
        const uint4 A = shfl_u4(rec, gl0), v1 = shfl_u4(rec, gl0 + 1), v2 = shfl_u4(rec, gl0 + 2);
        TriSetup t;
        if (!tri_setup_snapped(t, (int)A.x, (int)A.y, __uint_as_float(A.z), (int)v1.x, (int)v1.y, __uint_as_float(v1.z),
                               (int)v2.x, (int)v2.y, __uint_as_float(v2.z), a.cull))
            continue;
        int px0 = floordiv_subpix(t.minX - kSubpix / 2 + kSubpix - 1), px1 = floordiv_subpix(t.maxX - kSubpix / 2);
        int py0 = floordiv_subpix(t.minY - kSubpix / 2 + kSubpix - 1), py1 = floordiv_subpix(t.maxY - kSubpix / 2);
        px0 = max(px0, 0); py0 = max(py0, 0); px1 = min(px1, W - 1); py1 = min(py1, H - 1);
        // A triangle that COVERS more than kHugeArea pixels (a cell the camera has almost walked into: its vertices project far
        // outside the frame) is not for 16 lanes.  Its rows go, in blocks of kHugeRows, to the list k_mesh_raster_huge deals over
        // the whole chip, a wave per block.
        if ((t.area2 >> 17) > (i64)kHugeArea && (i64)(px1 - px0 + 1) * (py1 - py0 + 1) > (i64)kHugeArea && px1 >= px0 && py1 >= py0) {
            const uint32_t nblk = (uint32_t)(py1 - py0) / kHugeRows + 1u;
            uint32_t base = 0;
            if (MODE == 0 && sub == 0) base = atomicAdd(&a.hugeq[2 * kHugeCap], nblk);
            base = __shfl(base, (int)(threadIdx.x & 63u) & ~15);
            if (MODE == 0 && base + nblk <= (uint32_t)kHugeCap) {
                for (uint32_t b = sub; b < nblk; b += 16)
                    *(uint2*)(a.hugeq + 2 * (size_t)(base + b)) = make_uint2(id, e.y | (b << 8));
                if (sub == 0) ((uint2*)q)[k].y = e.y | 0x80000000u;  // (the second pass skips it here and finds it in the huge list)
                continue;
            }
            // list full: the triangle stays here, and the entries from `base` on were never written (every later push fails too)
            if (MODE == 0 && sub == 0 && base < (uint32_t)kHugeCap) atomicMax(&a.hugeq[2 * kHugeCap + 1], (uint32_t)kHugeCap - base);
        }
        const FragOut fo = frag_out(a, slot, eye);
        // (r04) what is constant for the triangle -- 1 / f32(area2), the colour planes' conversions are the fragment's -- and the edge
        // values advanced by exact integer steps along a row (16 pixels: w_k -= 16 * 256 dy_k) instead of six 32 x 32 -> 64
        // products per pixel centre; the same integers, the same f32 operations as tri_sample / tri_weights, so the same bits
        const bool small = t.area2 < 0x7FFFFFFFll;                      // 0 <= w_k <= area2 inside: every value fits int32
        const float ra = rcp_exact(i64_to_f32(t.area2, small));
        const i64 st0 = (i64)t.dy0 * (16 * kSubpix), st1 = (i64)t.dy1 * (16 * kSubpix), st2 = (i64)t.dy2 * (16 * kSubpix);
        if (py1 - py0 >= 3) {
            // Four rows or more: a ROW per lane, each lane walking its own short span pixel by pixel -- the rubber sheet between a
            // near object and the background under a pose with vertical parallax is a diagonal sliver hundreds of rows tall and two
            // or three pixels wide per row, and 16 lanes striding along each of its rows in turn did one row's set-up per two pixels
            const i64 s0 = (i64)t.dy0 * kSubpix, s1 = (i64)t.dy1 * kSubpix, s2 = (i64)t.dy2 * kSubpix;
            for (int pyb = py0; pyb <= py1; pyb += 16) {
                const int py = pyb + sub;
                int lo, hi;
                if (py > py1 || !tri_row_range(t, py, px0, px1, lo, hi)) continue;
                const int Xc = lo * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
                i64 w0 = mul64(t.dx0, Yc - t.by0) - mul64(t.dy0, Xc - t.bx0);
                i64 w1 = mul64(t.dx1, Yc - t.by1) - mul64(t.dy1, Xc - t.bx1);
                i64 w2 = mul64(t.dx2, Yc - t.by2) - mul64(t.dy2, Xc - t.bx2);
                for (int px = lo; px <= hi; ++px, w0 -= s0, w1 -= s1, w2 -= s2) {
                    if (!(edge_in(w0, t.dx0, t.dy0) && edge_in(w1, t.dx1, t.dy1) && edge_in(w2, t.dx2, t.dy2))) continue;
                    const float f0 = i64_to_f32(w0, small), f1 = i64_to_f32(w1, small), f2 = i64_to_f32(w2, small);
                    const float q0 = (f0 * ra) * t.iz0, q1 = (f1 * ra) * t.iz1, q2 = (f2 * ra) * t.iz2;
                    mesh_global_fragment<MODE>(a, slot, eye, fo, pd, W, px, py, q0, q1, q2, A.w, v1.w, v2.w, id);
                }
            }
            continue;
        }
        for (int py = py0; py <= py1; ++py) {
            int lo, hi;
            if (!tri_row_range(t, py, px0, px1, lo, hi)) continue;
            int px = lo + sub;
            if (px > hi) continue;
            const int Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
            i64 w0 = mul64(t.dx0, Yc - t.by0) - mul64(t.dy0, Xc - t.bx0);
            i64 w1 = mul64(t.dx1, Yc - t.by1) - mul64(t.dy1, Xc - t.bx1);
            i64 w2 = mul64(t.dx2, Yc - t.by2) - mul64(t.dy2, Xc - t.bx2);
            for (; px <= hi; px += 16, w0 -= st0, w1 -= st1, w2 -= st2) {
                if (!(edge_in(w0, t.dx0, t.dy0) && edge_in(w1, t.dx1, t.dy1) && edge_in(w2, t.dx2, t.dy2))) continue;
                float f0, f1, f2;
                if (__ballot(!small) == 0ull) { f0 = (float)(int)w0; f1 = (float)(int)w1; f2 = (float)(int)w2; }     // (as tri_weights)
                else { f0 = i64_to_f32(w0, small); f1 = i64_to_f32(w1, small); f2 = i64_to_f32(w2, small); }
                const float q0 = (f0 * ra) * t.iz0, q1 = (f1 * ra) * t.iz1, q2 = (f2 * ra) * t.iz2;
                mesh_global_fragment<MODE>(a, slot, eye, fo, pd, W, px, py, q0, q1, q2, A.w, v1.w, v2.w, id);
            }
        }
    }
    if (MODE == 0) pending_settle(a, pd);
}

// The row blocks of the huge triangles: a wave per entry, its 64 lanes along the rows (edge values advanced by 64 pixels per step).
// Same set-up, same integers, same f32 operations as k_mesh_raster_queue.
template <int MODE>
__device__ __forceinline__ void mesh_huge_walk(const RenderArgs& a)
{
    const int W = a.W, H = a.H;
    const uint32_t cnt = a.hugeq[2 * kHugeCap];
    const uint32_t lim = (uint32_t)kHugeCap - a.hugeq[2 * kHugeCap + 1];              // (what did not fit was rasterised by the queue kernel)
    const uint32_t total = cnt < lim ? cnt : lim;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    Pending pd = pending_none();
    for (uint32_t g = wave; g < total; g += nwaves) {
        const uint2 e = *(const uint2*)(a.hugeq + 2 * (size_t)g);
        const uint32_t id = e.x;
        const int eye = (int)(e.y & 1u), slot = (int)((e.y >> 1) & 0x7Fu), blk = (int)(e.y >> 8);
        if (MODE == 1 && a.tie_flag[slot] == 0u) continue;              // second pass: the frames with a marked pixel only
        const int pass = (int)(id >> 31), ci = (int)((id >> 16) & 0x7FFFu), cj = (int)(id & 0xFFFFu);
        const int vf = a.frame0 + slot;
        const FrameDev& vfp = a.fp[vf];
        const int vsel = lane < 3 ? lane : 0;                       // (as the queue walk: a vertex each by the wave's first three lanes)
        const uint4 rec = vertex_record_eye(a, vfp, vf, ci + (vsel == 1 || (vsel == 2 && pass == 0) ? 1 : 0),
                                            cj + (vsel == 2 || (vsel == 1 && pass != 0) ? 1 : 0), eye);
        const uint4 A = readlane_u4(rec, 0), v1 = readlane_u4(rec, 1), v2 = readlane_u4(rec, 2);
        TriSetup t;
        if (!tri_setup_snapped(t, (int)A.x, (int)A.y, __uint_as_float(A.z), (int)v1.x, (int)v1.y, __uint_as_float(v1.z),
                               (int)v2.x, (int)v2.y, __uint_as_float(v2.z), a.cull))
            continue;
        int px0 = floordiv_subpix(t.minX - kSubpix / 2 + kSubpix - 1), px1 = floordiv_subpix(t.maxX - kSubpix / 2);
        int py0 = floordiv_subpix(t.minY - kSubpix / 2 + kSubpix - 1), py1 = floordiv_subpix(t.maxY - kSubpix / 2);
        px0 = max(px0, 0); py0 = max(py0, 0); px1 = min(px1, W - 1); py1 = min(py1, H - 1);
        const int ya = py0 + blk * kHugeRows, yb = min(ya + kHugeRows - 1, py1);
        const FragOut fo = frag_out(a, slot, eye);
        const bool small = t.area2 < 0x7FFFFFFFll;
        const float ra = rcp_exact(i64_to_f32(t.area2, small));
        const i64 st0 = (i64)t.dy0 * (64 * kSubpix), st1 = (i64)t.dy1 * (64 * kSubpix), st2 = (i64)t.dy2 * (64 * kSubpix);
        for (int py = ya; py <= yb; ++py) {
            int lo, hi;
            if (!tri_row_range(t, py, px0, px1, lo, hi)) continue;
            int px = lo + lane;
            if (px > hi) continue;
            const int Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
            i64 w0 = mul64(t.dx0, Yc - t.by0) - mul64(t.dy0, Xc - t.bx0);
            i64 w1 = mul64(t.dx1, Yc - t.by1) - mul64(t.dy1, Xc - t.bx1);
            i64 w2 = mul64(t.dx2, Yc - t.by2) - mul64(t.dy2, Xc - t.bx2);
            for (; px <= hi; px += 64, w0 -= st0, w1 -= st1, w2 -= st2) {
                if (!(edge_in(w0, t.dx0, t.dy0) && edge_in(w1, t.dx1, t.dy1) && edge_in(w2, t.dx2, t.dy2))) continue;
                const float f0 = i64_to_f32(w0, small), f1 = i64_to_f32(w1, small), f2 = i64_to_f32(w2, small);
                const float q0 = (f0 * ra) * t.iz0, q1 = (f1 * ra) * t.iz1, q2 = (f2 * ra) * t.iz2;
                mesh_global_fragment<MODE>(a, slot, eye, fo, pd, W, px, py, q0, q1, q2, A.w, v1.w, v2.w, id);
            }
        }
    }
    if (MODE == 0) pending_settle(a, pd);
}

__global__ void __launch_bounds__(256) k_mesh_raster_queue(RenderArgs a, int nseg)
{
    extern __shared__ uint32_t cpre[];                    // (blocks + 1 + threads words: launch_mesh_raster_general)
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (r05 diagnosis, tuning build)
    queue_prefix_lds(a, nseg, cpre);
    mesh_queue_walk<0>(a, nseg, cpre);
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__global__ void __launch_bounds__(256) k_mesh_raster_huge(RenderArgs a) { mesh_huge_walk<0>(a); }

// The second pass, ONE launch of a fixed number of workgroups: for the frames in which the first pass marked a pixel (normally none:
// the launch is over after `nframes` loads per workgroup) every triangle near a marked tile posts draw id | colour into the side
// words of the marked pixels at the word's depth -- the cells again (k_mesh_raster_small's code), then the first pass's own lists
// of queued and huge triangles.  No order between the three: all they do is atomic minima.
template <int FLAGS, bool CONV>
__global__ void __launch_bounds__(kCellTPB) k_mesh_tie_pass(RenderArgs a, int nframes, int nseg)
{
    static_assert(kCellTPB == kConvTPB, "one workgroup size for both cell walks");
    __shared__ uint4 sv[2][2][kCellTPB];
    __shared__ uint32_t glist[2 * kConvTPB];
    __shared__ uint32_t gcount;
    extern __shared__ uint32_t cpre[];
    uint32_t any = 0;
    for (int fr = 0; fr < nframes; ++fr) any |= a.tie_flag[fr];
    if (!any) return;                                                          // (uniform)
    Pending pds[2] = {pending_none(), pending_none()};                         // (unused in this mode)
    const uint32_t nbx = (uint32_t)((a.W - 1 + kCellsWG - 1) / kCellsWG), nblk = nbx * (uint32_t)(a.H - 1);      // (here: a row of cells each)
    for (int fr = 0; fr < nframes; ++fr) {
        if (a.tie_flag[fr] == 0u) continue;                                    // (workgroup uniform)
        for (uint32_t v = blockIdx.x; v < nblk; v += gridDim.x) {
            const int bx = (int)(v % nbx), i = (int)(v / nbx);
            stage_vertex_row<CONV>(a, fr, i, bx * kCellsWG, sv, 0);
            stage_vertex_row<CONV>(a, fr, i + 1, bx * kCellsWG, sv, 1);
            __syncthreads();
            // (the cells classified as the first pass classified them: k_mesh_raster_conv draws spans itself that k_mesh_raster_small
            //  would have queued)
            if (CONV) mesh_raster_conv_block<FLAGS, 1>(a, fr, bx, i, sv, 0, glist, gcount, pds[0]);
            else mesh_raster_small_block<FLAGS, 1>(a, fr, bx, i, sv, 0, pds);
            __syncthreads();
        }
    }
    queue_prefix_lds(a, nseg, cpre);
    mesh_queue_walk<1>(a, nseg, cpre);
    mesh_huge_walk<1>(a);
}

// The rasteriser's counters of a launch set: the queue's segment counters, the tie flags and tile bits of its frames, the huge
// list's two counters.  (Until r04 the vertex pass zeroed them on its way.)
__global__ void __launch_bounds__(256) k_mesh_queue_reset(RenderArgs a, int n)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, nt = gridDim.x * 256u;
    const uint32_t nseg = (uint32_t)n * (uint32_t)a.H, ntile = (uint32_t)n * 2u * (uint32_t)a.tie_words;
    if (MDVT_DEBUG_SKIP(a) & 256) { for (uint32_t k = t; k < nseg; k += nt) atomicExch(&a.bigq_count[k], 0u); }      // (r05 diagnosis, tuning build)
    else for (uint32_t k = t; k < nseg; k += nt) a.bigq_count[k] = 0u;
    for (uint32_t k = t; k <= ((nseg + (1u << a.bigq_shift) - 1u) >> a.bigq_shift); k += nt) a.bigq_coarse[k] = 0u;
    for (uint32_t k = t; k < ntile; k += nt) a.tie_tiles[k] = 0u;
    if (t < (uint32_t)n) a.tie_flag[t] = 0u;
    if (t < 2u) a.hugeq[2 * (size_t)kHugeCap + t] = 0u;
    if (a.vlist_count && t < (uint32_t)n) a.vlist_count[t] = 0u;             // (the edge-point splat's list counters: one launch fewer)
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

hipError_t launch_mesh_raster_general(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    const int nseg = plan.n * a.H;
    a.bigq_coarse = a.bigq_count + nseg;                 // (nseg + 1 words follow the segment counters: mdvt_api.hip)
    a.bigq_shift = queue_shift_for(nseg);
    const int ncoarse = (nseg + (1 << a.bigq_shift) - 1) >> a.bigq_shift;
    const dim3 grid_c(cell_blocks(a.W, a.H, rows_wg(plan.remove_edges ? 2 : 0, false)), 1, plan.n);
    hipError_t e;
    {
        const uint32_t words = (uint32_t)nseg + (uint32_t)plan.n * 2u * (uint32_t)a.tie_words;
        hipLaunchKernelGGL(k_mesh_queue_reset, dim3(words / 1024u + 1u), dim3(256), 0, s, a, plan.n);
    }
    const bool conv = plan.conv_raster && tuning_env(TUNE_RASTER_CONV_OFF) == nullptr;
    if (conv) {
        // frames with nothing but a toe-in (every frame of the launch: plan.conv_raster): scanline intervals instead of triangles
        const dim3 grid_v(cell_blocks(a.W, a.H, rows_wg(plan.remove_edges ? 2 : 0, true)), 1, plan.n);
        if (plan.remove_edges) hipLaunchKernelGGL((k_mesh_raster_conv<2>), grid_v, dim3(kConvTPB), 0, s, a);
        else hipLaunchKernelGGL((k_mesh_raster_conv<0>), grid_v, dim3(kConvTPB), 0, s, a);
    } else if (plan.remove_edges) hipLaunchKernelGGL((k_mesh_raster_small<2>), grid_c, dim3(kCellTPB), 0, s, a);
    else hipLaunchKernelGGL((k_mesh_raster_small<0>), grid_c, dim3(kCellTPB), 0, s, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // (banks, mdvt_render_stereo_batch: the next launch set starts once this one's cell walk is through)
    if (plan.after_vertices && (e = hipEventRecord(plan.after_vertices, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_mesh_raster_queue, dim3(2048), dim3(256), (size_t)(ncoarse + 1 + 256) * sizeof(uint32_t), s, a, nseg);
    hipLaunchKernelGGL(k_mesh_raster_huge, dim3(2048), dim3(256), 0, s, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (tuning_env(TUNE_QUEUE_DUMP)) {       // tuning hook: queued (large) triangles and marked frames of this launch set on stderr
        uint32_t total = 0, marked = 0;
        std::vector<uint32_t> coarse((size_t)ncoarse);
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(coarse.data(), a.bigq_coarse, coarse.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            for (uint32_t v : coarse) total += v;
            for (int f = 0; f < plan.n; ++f) { uint32_t v = 0; if (hipMemcpy(&v, a.tie_flag + f, 4, hipMemcpyDeviceToHost) == hipSuccess) marked += v; }
            fprintf(stderr, "queued triangles: %u in %d frames (%d x %d), %u frames with pixels marked as tied\n", total, plan.n, a.W, a.H, marked);
        }
    }
    // the frames in which a pixel was marked as an exact depth tie between colours: those pixels settled by draw id (mdvt_device.h)
    const size_t tie_lds = (size_t)(ncoarse + 1 + kCellTPB) * sizeof(uint32_t);
    if (conv) {
        if (plan.remove_edges) hipLaunchKernelGGL((k_mesh_tie_pass<2, true>), dim3(2048), dim3(kCellTPB), tie_lds, s, a, plan.n, nseg);
        else hipLaunchKernelGGL((k_mesh_tie_pass<0, true>), dim3(2048), dim3(kCellTPB), tie_lds, s, a, plan.n, nseg);
    } else if (plan.remove_edges) hipLaunchKernelGGL((k_mesh_tie_pass<2, false>), dim3(2048), dim3(kCellTPB), tie_lds, s, a, plan.n, nseg);
    else hipLaunchKernelGGL((k_mesh_tie_pass<0, false>), dim3(2048), dim3(kCellTPB), tie_lds, s, a, plan.n, nseg);
    return hipGetLastError();
}

}  // namespace MDVT_GRID
}  // namespace mdvt
