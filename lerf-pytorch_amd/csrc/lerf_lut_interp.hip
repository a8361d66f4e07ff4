// A1 as a kernel of its own: one (LUT, pattern) simplex pass = FourSimplexInterpFaster (resample/eval_lut_sr.py:24-470),
// behind the reference's own signature (lerf_lut_interp, include/lerf_hip.h).  gfx950 only.
//
// lut_interp_lds_kernel -- the production form for the shipped interval (4): PERSISTENT 1024-thread workgroups, one per CU,
//   each with one byte plane of the LUT (83 521 B) resident in LDS for its whole life.  A LUT with oC = 3 is three planes:
//   workgroup b serves plane b % 3 and walks the tile sequence of that plane, so a tile is visited by three workgroups, each
//   storing one output channel -- every store instruction of a wave writes 512 contiguous bytes of one float64 plane row
//   (three byte planes do not fit the 160 KB together; the dword-packed form of the fused kernel needs its binning machinery).
//   Per tile (64 x 64 positions x C channels): the pixels (+ the pattern's reach, <= 3) are staged as BYTES in LDS in the
//   image's own memory order (float32 -> uint8 once per pixel, 16-byte loads, dword LDS stores; the next tile's loads are
//   issued before the current tile's walks and land in registers meanwhile); a wave takes a row of 64 positions -- along the
//   axis on which the OUTPUT plane is contiguous, so np.rot90(out, rot) of the reference's epilogue (:464-468) keeps the
//   stores dense for every rot -- and runs two rows interleaved: d16_hi pixel reads, the 7-instruction sort, 5 byte gathers,
//   Abel sum, value = numerator / 16 converted in the store.  `accumulate` adds into the plane instead (the call sites'
//   `pred += FourSimplexInterpFaster(...)`, :555/:564, without a second pass over 50 / 150 MB).
// lut_interp_kernel / lut_interp_any_kernel -- the direct forms (LUT gathered from L1 / L2): small launches, other
//   intervals, patterns reaching further than 3 pixels, more than 4 channels.
#include "lerf_kernels.h"

namespace lerf {

// ---------------------------------------------------------------------------
// direct kernels
// ---------------------------------------------------------------------------
// The image operand is uint8 or float32 (the call sites hand over float32 arrays of integer values, resample/eval_lut_sr.py:
// 549-553; anything else is rounded half-to-even and clipped to 0..255, what `.round().clamp(0, 255)` gave the uint8 kernels
// before); the result goes out through a plane with SIGNED element strides -- int16 numerators, or float32 / float64 VALUES
// (numerator / q, exact: q is a power of two) -- so the caller's np.rot90(out, rot, [1, 2]) (:464-468) and the division (:469)
// are part of the store.  16 x 16 positions per workgroup: rows of 16 elements in either orientation of the result.
template <typename TIN>
__device__ __forceinline__ int pixel_value(const TIN* __restrict__ p);
template <>
__device__ __forceinline__ int pixel_value<uint8_t>(const uint8_t* __restrict__ p) { return (int)*p; }
template <>
__device__ __forceinline__ int pixel_value<float>(const float* __restrict__ p) {
    return (int)__builtin_rintf(fminf(fmaxf(*p, 0.0f), 255.0f));
}
template <typename TOUT, bool ACC>
__device__ __forceinline__ void store_interp(TOUT* __restrict__ o, int acc, float inv_q);
template <>
__device__ __forceinline__ void store_interp<int16_t, false>(int16_t* __restrict__ o, int acc, float) { *o = (int16_t)acc; }
template <>
__device__ __forceinline__ void store_interp<float, false>(float* __restrict__ o, int acc, float inv_q) { *o = (float)acc * inv_q; }
template <>
__device__ __forceinline__ void store_interp<double, false>(double* __restrict__ o, int acc, float inv_q) { *o = (double)acc * (double)inv_q; }
template <>
__device__ __forceinline__ void store_interp<int16_t, true>(int16_t* __restrict__ o, int acc, float) { *o = (int16_t)(*o + acc); }
template <>
__device__ __forceinline__ void store_interp<float, true>(float* __restrict__ o, int acc, float inv_q) { *o = *o + (float)acc * inv_q; }
template <>
__device__ __forceinline__ void store_interp<double, true>(double* __restrict__ o, int acc, float inv_q) { *o = *o + (double)acc * (double)inv_q; }

template <int OC, typename TIN, typename TOUT, bool ACC>
__global__ void __launch_bounds__(256)
lut_interp_kernel(const TIN* __restrict__ img, int64_t sy, int64_t sx, int64_t sc,
                  int img_h, int img_w, int C, int h, int w, Offsets4 off,
                  const int8_t* __restrict__ lut, TOUT* __restrict__ out, int64_t oy, int64_t ox, int64_t oc_stride, float inv_q) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15);
    int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    int c = blockIdx.z;
    if (x >= w || y >= h) return;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int yy = clampi(y + off.dy[k], 0, img_h - 1);
        int xx = clampi(x + off.dx[k], 0, img_w - 1);
        v[k] = pixel_value<TIN>(img + (yy * sy + xx * sx + c * sc));
    }
    SimplexPath p = simplex_path(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) {
        int acc = 0;
#pragma unroll
        for (int n = 0; n < 5; ++n) acc += p.w[n] * (int)lut[p.idx[n] * OC + oc];
        store_interp<TOUT, ACC>(out + (((int64_t)c * OC + oc) * oc_stride + (int64_t)y * oy + (int64_t)x * ox), acc, inv_q);
    }
}

// the same pass for any sampling interval (resample/eval_lut_sr.py:27-28: q = 2^interval, L = 2^(8-interval) + 1);
// the shipped LUTs and every fused path use interval 4, this one serves the function mirror for the others
template <int OC, typename TIN, typename TOUT, bool ACC>
__global__ void __launch_bounds__(256)
lut_interp_any_kernel(const TIN* __restrict__ img, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C, int h,
                      int w, Offsets4 off, const int8_t* __restrict__ lut, int interval, TOUT* __restrict__ out, int64_t oy, int64_t ox,
                      int64_t oc_stride, float inv_q) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15);
    int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    int c = blockIdx.z;
    if (x >= w || y >= h) return;
    const int q = 1 << interval, L = (1 << (8 - interval)) + 1;
    const int stride[4] = {L * L * L, L * L, L, 1};
    unsigned key[4];
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int yy = clampi(y + off.dy[k], 0, img_h - 1);
        int xx = clampi(x + off.dx[k], 0, img_w - 1);
        const int v = pixel_value<TIN>(img + (yy * sy + xx * sx + c * sc));
        idx += (v >> interval) * stride[k];
        key[k] = ((unsigned)(v & (q - 1)) << 24) | (unsigned)stride[k];      // L^3 <= 129^3 < 2^24
    }
    ce_desc(key[0], key[1]);
    ce_desc(key[2], key[3]);
    ce_desc(key[0], key[2]);
    ce_desc(key[1], key[3]);
    ce_desc(key[1], key[2]);
    int f[5], id[5];
    id[0] = idx;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        f[n] = (int)(key[n] >> 24);
        id[n + 1] = id[n] + (int)(key[n] & 0xFFFFFFu);
    }
    f[4] = 0;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) {
        int acc = (q - f[0]) * (int)lut[(int64_t)id[0] * OC + oc];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc += (f[n] - f[n + 1]) * (int)lut[(int64_t)id[n + 1] * OC + oc];
        store_interp<TOUT, ACC>(out + (((int64_t)c * OC + oc) * oc_stride + (int64_t)y * oy + (int64_t)x * ox), acc, inv_q);
    }
}

// ---------------------------------------------------------------------------
// the LDS-resident form
// ---------------------------------------------------------------------------
namespace {

constexpr int LI_NT = 1024;                 // threads per workgroup: 16 waves, 4 per SIMD
constexpr int LI_NW = LI_NT / 64;
constexpr int LI_NCW = 12;                  // waves 0..11 walk the current tile,
constexpr int LI_NLW = LI_NW - LI_NCW;      // waves 12..15 (one per SIMD) stage the next one meanwhile
constexpr int LI_NI = 4;                    // wave-rows a compute wave runs interleaved (20 gathers in flight)
constexpr int LI_TL = 64;                   // tile extent along the lane axis
constexpr int LI_REACH = 3;                 // largest extent of a pattern the LDS tile provides for (s d y: 2, c t: 3)
constexpr int LI_LH = LI_TL + LI_REACH;
constexpr int LI_CMAX = 4;
constexpr int LI_ENTRIES = kL * kL * kL * kL;                 // 83 521
constexpr int LI_LUT_BYTES = (LI_ENTRIES + 63) / 64 * 64;      // 83 584
constexpr int LI_ALL = kStrideA + kStrideB + kStrideC + kStrideD;                   // vertex 4 - vertex 0

// TO = tile extent along the other axis (64: fewest halo pixels; 32: twice the tiles, for launches of few tiles per workgroup)
template <int TO>
struct LiDims {
    static constexpr int OH = TO + LI_REACH;
    // pixel tile: staged rows padded to a multiple of 4 bytes with an odd number of dwords (<= 7 bytes of padding per row)
    static constexpr int PIX_BYTES = (LI_CMAX * LI_LH * OH + 7 * LI_LH * LI_CMAX + 15) / 16 * 16;
    static constexpr int GROUPS_MAX = LI_CMAX * LI_LH * ((OH + 3) / 4) > LI_CMAX * OH * ((LI_LH + 3) / 4)
                                          ? LI_CMAX * LI_LH * ((OH + 3) / 4) : LI_CMAX * OH * ((LI_LH + 3) / 4);   // 4-pixel groups of a tile
    static constexpr int LDS_BYTES = LI_LUT_BYTES + 2 * PIX_BYTES;
    // tiles whose loads a loader wave keeps in flight: 2 x 10 x 16 bytes per thread for 64 x 32 tiles; a 64 x 64 tile alone is 18 x 16
    static constexpr int DEPTH = TO == 32 ? 2 : 1;
};

enum { LI_LAYOUT_GENERIC = 0, LI_LAYOUT_HWC = 1, LI_LAYOUT_PLANAR = 2 };

struct LiArgs {
    const void* img; int64_t sy, sx, sc;     // element strides of the image operand [C][img_h][img_w]
    int64_t max_off;                         // largest element offset inside the operand
    int img_h, img_w, C, h, w;
    const int8_t* lut;
    void* out; int64_t ocs;                  // out plane (c * OC + oc) starts at out + (c * OC + oc) * ocs
    int64_t ol, oo;                          // output strides along the lane axis / the other axis
    int lane_is_y;                           // lanes run along y (the output plane is contiguous along y: rot 1, 3)
    int tiles_x, tiles_y, tile_h, tile_w;    // tile grid; tile extent in rows / columns (64 x TO or TO x 64)
    int miny, minx, maxx;                    // smallest dy, dx and largest dx of the pattern
    int th_y, th_x;                          // staged rows / columns of a tile (tile + reach)
    int layout;                              // LI_LAYOUT_*: how the tile is laid out in LDS (= memory order of the operand)
    int pitch, cs, xs;                       // LDS byte strides: rows, channels, columns
    int rows, groups;                        // staged rows (th_y or th_y C), 4-byte groups per staged row
    unsigned groups_magic, thy_magic;        // floor(i / groups) = (i * magic) >> 20, floor(r / th_y) likewise
    int koff[4];                             // LDS byte offset of pattern pixel k from the position's own tile cell
#ifdef LERF_LI_STAMPS
    unsigned long long* stamps;              // diagnostic build: 32 s_memrealtime stamps (100 MHz) per workgroup
#endif
};
#ifdef LERF_LI_STAMPS
unsigned long long* g_li_stamps = nullptr;
#define LI_STAMP(k) do { if (tid == 0 && A.stamps && (k) < 32) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); \
                         A.stamps[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
#define LI_STAMP_L(k) do { if (tid == LI_NCW * 64 && A.stamps && (k) < 32) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); \
                         A.stamps[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
#else
#define LI_STAMP(k) do {} while (0)
#define LI_STAMP_L(k) do {} while (0)
#endif

typedef __attribute__((address_space(3))) const int8_t li_lds_i8_t;
typedef float li_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t li_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// ds_read_u8_d16_hi: the byte lands in bits 16..23 (v << 16 for free), so key = one v_and_or_b32 and the LUT digit one shift
__device__ __forceinline__ uint32_t li_pixel_hi(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
__device__ __forceinline__ unsigned li_key(uint32_t r, unsigned stride) {
    unsigned k;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(k) : "v"(r), "s"(0x000F0000u), "v"(stride));
    return k;
}
__device__ __forceinline__ unsigned li_max3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned li_med3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned li_min3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// acc + d.lo16 * key.hi16 (signed): one Abel term, the LSB read in place from the sorted key's high half
__device__ __forceinline__ int li_mad_keyhi(int d, unsigned key, int acc) {
    asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc) : "v"(d), "v"(key));
    return acc;
}

struct LiWalk {
    uint32_t i0, i1, i2, i3m;        // LDS addresses of vertices 0, 1, 2 and (vertex 3 - LI_ALL); vertex 4 = i0 + LI_ALL
    unsigned k0, k1, k2, k3;         // sorted keys (LSB << 16 | stride)
};
// ra..rd: the four pattern pixels as li_pixel_hi() returned them; lut_a: LDS address of the plane
__device__ __forceinline__ LiWalk li_walk(uint32_t lut_a, uint32_t ra, uint32_t rb, uint32_t rc, uint32_t rd) {
    LiWalk W;
    const unsigned ka = li_key(ra, kStrideA), kb = li_key(rb, kStrideB), kc = li_key(rc, kStrideC), kd = li_key(rd, kStrideD);
    const unsigned t = __umul24(__umul24(__umul24(ra >> 20, (unsigned)kL) + (rb >> 20), (unsigned)kL) + (rc >> 20), (unsigned)kL) + (rd >> 20);
    W.i0 = lut_a + t;
    const unsigned m = li_max3(ka, kb, kc), e = li_med3(ka, kb, kc), n = li_min3(ka, kb, kc);
    W.k0 = m > kd ? m : kd;
    W.k1 = li_med3(m, e, kd);
    W.k2 = li_med3(e, n, kd);
    W.k3 = n < kd ? n : kd;
    W.i1 = W.i0 + (W.k0 & 0xFFFFu);
    W.i2 = W.i1 + (W.k1 & 0xFFFFu);
    W.i3m = W.i0 - (W.k3 & 0xFFFFu);
    return W;
}
struct LiEntries { int e0, e1, e2, e3, e4; };
__device__ __forceinline__ LiEntries li_gather(const LiWalk& W) {
    LiEntries E;
    E.e0 = (int)((li_lds_i8_t*)W.i0)[0];
    E.e1 = (int)((li_lds_i8_t*)W.i1)[0];
    E.e2 = (int)((li_lds_i8_t*)W.i2)[0];
    E.e3 = (int)((li_lds_i8_t*)W.i3m)[LI_ALL];
    E.e4 = (int)((li_lds_i8_t*)W.i0)[LI_ALL];
    return E;
}
// sum_n w_n P_n = 16 P_0 + sum_n f_n (P_{n+1} - P_n)   (w_0 = 16 - f_0, w_n = f_{n-1} - f_n, w_4 = f_3)
__device__ __forceinline__ int li_numerator(const LiWalk& W, const LiEntries& E) {
    int acc = kQ * E.e0;
    acc = li_mad_keyhi(E.e1 - E.e0, W.k0, acc);
    acc = li_mad_keyhi(E.e2 - E.e1, W.k1, acc);
    acc = li_mad_keyhi(E.e3 - E.e2, W.k2, acc);
    acc = li_mad_keyhi(E.e4 - E.e3, W.k3, acc);
    return acc;
}

#ifdef LERF_LI_NT_STORE
#define LI_ST(v, o) __builtin_nontemporal_store(v, o)
#else
#define LI_ST(v, o) (*(o) = (v))
#endif
template <typename TOUT, bool ACC>
__device__ __forceinline__ void li_store(TOUT* __restrict__ o, int acc) {
    if constexpr (sizeof(TOUT) == 2) {
        *o = ACC ? (int16_t)(*o + acc) : (int16_t)acc;
    } else if constexpr (sizeof(TOUT) == 4) {
        const float v = (float)acc * 0.0625f;
        if constexpr (ACC) *o = *o + v; else LI_ST(v, o);
    } else {
        const double v = (double)((float)acc * 0.0625f);          // |acc| <= 2 032: exact in float32
        if constexpr (ACC) *o = *o + v; else LI_ST(v, o);
    }
}

// four pixels -> four bytes, rounded half-to-even and clipped like pixel_value<float>
__device__ __forceinline__ uint32_t li_pack4(const li_v4f& v) {
    const int a = (int)__builtin_rintf(fminf(fmaxf(v.x, 0.0f), 255.0f)), b = (int)__builtin_rintf(fminf(fmaxf(v.y, 0.0f), 255.0f));
    const int c = (int)__builtin_rintf(fminf(fmaxf(v.z, 0.0f), 255.0f)), d = (int)__builtin_rintf(fminf(fmaxf(v.w, 0.0f), 255.0f));
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}

// What a thread holds of one 4-pixel group between the load and the LDS store.  The loads are inline assembly: issued
// where they stand (the compiler sinks plain loads to their use, behind the whole walk loop) and waited for by li_landed().
template <typename TIN> struct LiGroup;
template <> struct LiGroup<float> {
    li_v4f v;
    __device__ __forceinline__ void load(const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p)); }
    __device__ __forceinline__ void landed() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ uint32_t bytes() const { return li_pack4(v); }
};
template <> struct LiGroup<uint8_t> {
    uint32_t v;
    __device__ __forceinline__ void load(const uint8_t* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p)); }
    __device__ __forceinline__ void landed() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ uint32_t bytes() const { return v; }
};
struct LiTile { int y0, x0, fast; };

__device__ __forceinline__ LiTile li_tile_of(const LiArgs& A, int t) {
    LiTile T;
    const int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
    T.y0 = ty * A.tile_h;
    T.x0 = tx * A.tile_w;
    // fast staging (16-byte loads in the operand's memory order): rows are clamped to the operand like the pass clamps its
    // coordinates; columns are not, so every column a position of the frame reads must exist -- columns beyond them hold
    // whatever follows in memory (read only by positions outside the frame) -- and the last group must end inside the operand
    const int xl = T.x0 + A.minx;
    const int x_need = min(T.x0 + A.tile_w, A.w) + A.maxx;                      // one past the last column a valid position reads
    const bool fast = A.layout != LI_LAYOUT_GENERIC && xl >= 0 && x_need <= A.img_w;
    T.fast = 0;
    if (fast) {
        // the last group of the last staged row may end beyond the operand (the frame's bottom right corner): such a tile
        // (fast = 2) loads its groups no further than the operand's last four elements and shifts the bytes back into place
        const int yl = min(max(T.y0 + A.miny + A.th_y - 1, 0), A.img_h - 1);      // last staged row, clamped
        const int64_t last = (A.layout == LI_LAYOUT_HWC ? 0 : (int64_t)(A.C - 1) * A.sc) + (int64_t)yl * A.sy + (int64_t)xl * A.sx +
                             (int64_t)A.groups * 4 - 1;
        T.fast = last <= A.max_off ? 1 : 2;
    }
    return T;
}

// element offset (32 bits: the host checks that the operand spans less than 2^31 elements) of 4-pixel group q of a fast tile;
// rows of an HWC tile are image rows (thy_magic = 0: c = 0), rows of a planar tile (channel, image row) pairs
__device__ __forceinline__ int li_group_src(const LiArgs& A, int yl, int xorg, int q) {
    const int r = (int)(((unsigned)q * A.groups_magic) >> 20), g = q - r * A.groups;
    const int c = (int)(((unsigned)r * A.thy_magic) >> 20), yy = r - c * A.th_y;
    const int gy = min(max(yl + yy, 0), A.img_h - 1);
    return xorg + c * (int)A.sc + gy * (int)A.sy + g * 4;
}

__device__ __forceinline__ void li_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int PENDING>
__device__ __forceinline__ void li_wait_loads_but() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PENDING) : "memory"); }

// fast tile -> LDS by NTHR threads (lt = this thread's index among them): every load in flight before the first conversion
template <typename TIN, int NTHR, int GROUPS_MAX>
struct LiStager {
    static constexpr int N = (GROUPS_MAX + NTHR - 1) / NTHR;
    LiGroup<TIN> pf[N];
    // lt arrives through an opaque asm at every tile: what depends on it (row / group pairs and their addresses) is
    // recomputed per tile (~10 instructions per group) instead of living in registers -- or in scratch -- across the tile loop
    __device__ __forceinline__ void issue(const LiArgs& A, const LiTile& T, int lt) {
        const int ng = A.rows * A.groups;
        const TIN* __restrict__ img = (const TIN*)A.img;
        const int yl = T.y0 + A.miny, xorg = (T.x0 + A.minx) * (int)A.sx;
        const int cap = T.fast == 1 ? 0x7FFFFFFF : (T.fast ? (int)A.max_off - 3 : 0);      // generic tiles: a harmless load
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // every lane loads (threads beyond the tile's last group repeat it): an assembly load under a branch leaves its
            // register undefined on the other path, which the compiler answers with a scratch slot and a wait per load
            const int q = min(lt + i * NTHR, ng - 1);
            pf[i].load(img + max(min(li_group_src(A, yl, xorg, q), cap), 0));    // (generic tiles may start left of the operand)
        }
    }
    // PENDING = loads this wave has issued after this tile's (deeper prefetch: the sets of the following tiles)
    template <int PENDING>
    __device__ __forceinline__ void commit(const LiArgs& A, const LiTile& T, uint8_t* pix, int lt) {
        const int ng = A.rows * A.groups;
        li_wait_loads_but<PENDING>();
        const int yl = T.y0 + A.miny, xorg = (T.x0 + A.minx) * (int)A.sx;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int q = lt + i * NTHR;
            pf[i].landed();
            uint32_t v = pf[i].bytes();
            if (T.fast == 2) {                             // wave-uniform, one tile per launch at most
                const int off = li_group_src(A, yl, xorg, min(q, ng - 1));
                const int sh = off - min(off, (int)A.max_off - 3);
                v = sh < 4 ? v >> (8 * sh) : 0u;           // the elements beyond the operand are read by no valid position
            }
            if (q < ng) {
                const int r = (int)(((unsigned)q * A.groups_magic) >> 20), g = q - r * A.groups;
                *reinterpret_cast<uint32_t*>(pix + r * A.pitch + g * 4) = v;
            }
        }
    }
};

// any strides, clamped coordinates: one pixel per thread and step (tiles on the frame's border, odd layouts)
template <typename TIN>
__device__ __forceinline__ void li_stage_generic(const LiArgs& A, const LiTile& T, uint8_t* pix, int lt, int nthr) {
    const TIN* __restrict__ img = (const TIN*)A.img;
    const int n = A.th_y * A.th_x * A.C;
#pragma unroll 4
    for (int i = lt; i < n; i += nthr) {
        int c, yy, xx;
        if (A.cs == 1) {                                   // rows of (column, channel)
            const int r = i / A.C;
            c = i - r * A.C;
            yy = r / A.th_x;
            xx = r - yy * A.th_x;
        } else {                                           // planes of rows
            const int r = i / A.th_x;
            xx = i - r * A.th_x;
            c = r / A.th_y;
            yy = r - c * A.th_y;
        }
        const int gy = clampi(T.y0 + A.miny + yy, 0, A.img_h - 1), gx = clampi(T.x0 + A.minx + xx, 0, A.img_w - 1);
        pix[c * A.cs + yy * A.pitch + xx * A.xs] = (uint8_t)pixel_value<TIN>(img + ((int64_t)gy * A.sy + (int64_t)gx * A.sx + (int64_t)c * A.sc));
    }
}

template <int OC, typename TIN, typename TOUT, bool ACC, int TO>
__global__ void __launch_bounds__(LI_NT)
lut_interp_lds_kernel(LiArgs A) {
    using D = LiDims<TO>;
    extern __shared__ __attribute__((aligned(16))) uint8_t li_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int plane = (int)blockIdx.x % OC, slot = (int)blockIdx.x / OC;
    const int nslots = ((int)gridDim.x - plane + OC - 1) / OC;
    const int ntiles = A.tiles_x * A.tiles_y;
    LI_STAMP(0);
    int t = slot;
    if (t >= ntiles) return;
    uint8_t* const pix0 = li_smem + LI_LUT_BYTES;
    LiTile T = li_tile_of(A, t);

    // ---- start-up, all 16 waves: the first tile's loads fly under the copy of this workgroup's LUT plane
    auto copy_lut = [&]() {
        uint4* L128 = reinterpret_cast<uint4*>(li_smem);
        constexpr int NQ = LI_ENTRIES / 16;                          // 5 220 whole 16-entry blocks, + 1 entry
        constexpr int NLQ = (NQ + LI_NT - 1) / LI_NT;                // 6 per thread
        if constexpr (OC == 1) {
            const uint4* s = reinterpret_cast<const uint4*>(A.lut);
            uint4 r[NLQ];
#pragma unroll
            for (int i = 0; i < NLQ; ++i) r[i] = s[min(tid + i * LI_NT, NQ - 1)];
#pragma unroll
            for (int i = 0; i < NLQ; ++i) if (tid + i * LI_NT < NQ) L128[tid + i * LI_NT] = r[i];
        } else {
            // 16 entries = 48 interleaved bytes -> 16 bytes of this plane
            const uint32_t* s = reinterpret_cast<const uint32_t*>(A.lut);
            const int sh = plane * 8;
#pragma unroll 2
            for (int i = 0; i < NLQ; ++i) {
                const int b = tid + i * LI_NT;
                if (b < NQ) {
                    uint32_t d[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) d[k] = s[12 * b + k];
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                    // entries 4 k .. 4 k + 3 of the block = dwords 3 k .. 3 k + 2
                        const uint64_t lo = (uint64_t)d[3 * k] | ((uint64_t)d[3 * k + 1] << 32);
                        const uint64_t hi = (uint64_t)d[3 * k + 1] | ((uint64_t)d[3 * k + 2] << 32);
                        const uint32_t b0 = (uint32_t)(lo >> sh) & 0xFFu, b1 = (uint32_t)(lo >> (sh + 24)) & 0xFFu;
                        const uint32_t b2 = (uint32_t)(hi >> (sh + 16)) & 0xFFu, b3 = (uint32_t)(hi >> (sh + 40)) & 0xFFu;
                        o[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    }
                    L128[b] = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        if (tid == 0) li_smem[LI_ENTRIES - 1] = (uint8_t)A.lut[(size_t)(LI_ENTRIES - 1) * OC + plane];
    };
    if (T.fast) {
        LiStager<TIN, LI_NT, D::GROUPS_MAX> S0;
        S0.issue(A, T, tid);
        copy_lut();
        LI_STAMP(1);
        S0.template commit<0>(A, T, pix0, tid);
    } else {
        copy_lut();
        LI_STAMP(1);
        li_stage_generic<TIN>(A, T, pix0, tid, LI_NT);
    }
    __syncthreads();
    LI_STAMP(2);
    const uint32_t lut_a = li_lds_addr(li_smem);

    // ---- the walks of one tile: wave-rows of 64 positions = (channel, other-axis index), LI_NI rows interleaved
    const int lstride = A.lane_is_y ? A.pitch : A.xs, ostride = A.lane_is_y ? A.xs : A.pitch;
    const int nl = A.lane_is_y ? A.h : A.w, no = A.lane_is_y ? A.w : A.h;
    TOUT* const outp = (TOUT*)A.out;
    const int nrows = A.C * TO;
    auto out_row = [&](int c, int o) { return outp + ((int64_t)(c * OC + plane) * A.ocs + (int64_t)o * A.oo); };
    auto compute = [&](const LiTile& Tc, const uint8_t* pix) {
        const int l0 = A.lane_is_y ? Tc.y0 : Tc.x0, o0 = A.lane_is_y ? Tc.x0 : Tc.y0;
        const bool lane_ok = l0 + lane < nl;
        const int lane_out = (int)((int64_t)(l0 + lane) * A.ol);
        const uint32_t pix_a = li_lds_addr(pix) + (uint32_t)(lane * lstride);
        const int ovalid = min(TO, no - o0);                               // rows of the tile inside the frame
#pragma unroll 1
        for (int r0 = wave; r0 < nrows; r0 += LI_NCW * LI_NI) {
            int c[LI_NI], o[LI_NI];
            bool ok[LI_NI];
            bool all = true;
#pragma unroll
            for (int j = 0; j < LI_NI; ++j) {
                const int r = r0 + j * LI_NCW;
                c[j] = r / TO;
                o[j] = r - c[j] * TO;
                ok[j] = r < nrows && o[j] < ovalid;                        // wave-uniform
                all = all && ok[j];
            }
            if (all) {
                uint32_t px[LI_NI][4];
#pragma unroll
                for (int j = 0; j < LI_NI; ++j) {
                    const uint32_t b = pix_a + (uint32_t)(c[j] * A.cs + o[j] * ostride);
#pragma unroll
                    for (int k = 0; k < 4; ++k) px[j][k] = li_pixel_hi(b + A.koff[k]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(px[0][0]), "+v"(px[0][1]), "+v"(px[0][2]), "+v"(px[0][3]), "+v"(px[1][0]), "+v"(px[1][1]), "+v"(px[1][2]), "+v"(px[1][3]),
                               "+v"(px[2][0]), "+v"(px[2][1]), "+v"(px[2][2]), "+v"(px[2][3]), "+v"(px[3][0]), "+v"(px[3][1]), "+v"(px[3][2]), "+v"(px[3][3]));
                LiWalk W[LI_NI];
                LiEntries E[LI_NI];
#pragma unroll
                for (int j = 0; j < LI_NI; ++j) {
#ifdef LERF_LI_NO_WALK
                    W[j].k0 = px[j][0]; W[j].k1 = px[j][1]; W[j].k2 = px[j][2]; W[j].k3 = px[j][3];
                    E[j].e0 = px[j][0]; E[j].e1 = px[j][1]; E[j].e2 = px[j][2]; E[j].e3 = px[j][3]; E[j].e4 = px[j][0];
#else
                    W[j] = li_walk(lut_a, px[j][0], px[j][1], px[j][2], px[j][3]);
                    E[j] = li_gather(W[j]);
                    __builtin_amdgcn_sched_barrier(0);                      // this walk's gathers fly under the next walk
#endif
                }
#pragma unroll
                for (int j = 0; j < LI_NI; ++j) {
                    const int nj = li_numerator(W[j], E[j]);
#ifdef LERF_LI_NO_STORE                                                     // ablation builds (tools/build_li_variant.sh): never the product
                    if (nj == 0x7FFFFFF && lane_ok)
#else
                    if (lane_ok)
#endif
                        li_store<TOUT, ACC>(out_row(c[j], o0 + o[j]) + lane_out, nj);
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < LI_NI; ++j) {
                    if (!ok[j]) continue;
                    const uint32_t b = pix_a + (uint32_t)(c[j] * A.cs + o[j] * ostride);
                    uint32_t p0 = li_pixel_hi(b + A.koff[0]), p1 = li_pixel_hi(b + A.koff[1]), p2 = li_pixel_hi(b + A.koff[2]), p3 = li_pixel_hi(b + A.koff[3]);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                    const LiWalk W1 = li_walk(lut_a, p0, p1, p2, p3);
                    const LiEntries E1 = li_gather(W1);
                    const int n1 = li_numerator(W1, E1);
#ifdef LERF_LI_NO_STORE
                    if (n1 == 0x7FFFFFF && lane_ok)
#else
                    if (lane_ok)
#endif
                        li_store<TOUT, ACC>(out_row(c[j], o0 + o[j]) + lane_out, n1);
                }
            }
        }
    };

    // ---- tile loop: the pixel tile is double-buffered; waves 0..11 walk tile t while waves 12..15 convert tile t + 1 (loaded
    // during the previous period) into the other buffer and then issue the loads of tile t + 2, which fly until the next
    // period.  Two loops with the same barrier count, so that the registers of the loads in flight are live in the loader's
    // loop only: the compiler must never spill or move the destination of an assembly load it cannot see landing.
    [[maybe_unused]] int stamp_k = 3;
    if (wave >= LI_NCW) {
#ifndef LERF_LI_NO_STAGE
        // D::DEPTH register sets: while tile t is walked, tile t + 1 is converted and the loads of tiles t + 2 .. t + 1 + DEPTH
        // are in flight (reads wait microseconds behind the launch's own stores: one tile of lead is not enough)
        int lt = tid - LI_NCW * 64;
        using Stager = LiStager<TIN, LI_NLW * 64, D::GROUPS_MAX>;
        Stager S[D::DEPTH];
        LiTile Ts[D::DEPTH];
        int cur = 0;
#pragma unroll
        for (int d = 0; d < D::DEPTH; ++d) {
            const int td = t + (1 + d) * nslots;
            Ts[d] = li_tile_of(A, td < ntiles ? td : t);
            asm volatile("" : "+v"(lt));
            S[d].issue(A, Ts[d], lt);                                   // (a tile of the generic kind loads nothing useful)
        }
        bool done = false;
#pragma unroll 1
        while (!done) {
#pragma unroll
            for (int d = 0; d < D::DEPTH; ++d) {
                if (t + nslots >= ntiles) { done = true; break; }
                uint8_t* nxt = pix0 + (cur ^ 1) * D::PIX_BYTES;
                asm volatile("" : "+v"(lt));
                if (Ts[d].fast) S[d].template commit<(D::DEPTH - 1) * Stager::N>(A, Ts[d], nxt, lt);
                else { li_wait_loads(); li_stage_generic<TIN>(A, Ts[d], nxt, lt, LI_NLW * 64); }
                const int tnn = t + (1 + D::DEPTH) * nslots;
                Ts[d] = li_tile_of(A, tnn < ntiles ? tnn : t);
                S[d].issue(A, Ts[d], lt);
                LI_STAMP_L(stamp_k + 1);
                stamp_k += 2;
                __syncthreads();
                cur ^= 1;
                t += nslots;
            }
        }
        li_wait_loads();                                                // nothing of this wave may be in flight at its end
#else
        while (t + nslots < ntiles) { __syncthreads(); t += nslots; }
#endif
    } else {
        int cur = 0;
#pragma unroll 1
        for (;;) {
            compute(T, pix0 + cur * D::PIX_BYTES);
            LI_STAMP(stamp_k);
            stamp_k += 2;
            if (t + nslots >= ntiles) break;
            __syncthreads();
            cur ^= 1;
            t += nslots;
            T = li_tile_of(A, t);
        }
    }
}

template <auto KERN>
int li_ensure_lds(int bytes) {
    static thread_local int done_dev[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LERF_ELAUNCH;
    if (!done_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return LERF_ELAUNCH;
        done_dev[dev] = 1;
    }
    return LERF_OK;
}

int li_cu_count() {
    static thread_local int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

template <int OC, typename TIN, typename TOUT, bool ACC, int TO>
int li_launch(const LiArgs& A, int grid, hipStream_t st) {
    constexpr auto K = lut_interp_lds_kernel<OC, TIN, TOUT, ACC, TO>;
    const int rc = li_ensure_lds<K>(LiDims<TO>::LDS_BYTES);
    if (rc != LERF_OK) return rc;
    hipLaunchKernelGGL(K, dim3(grid), dim3(LI_NT), LiDims<TO>::LDS_BYTES, st, A);
    return LERF_OK;
}

}  // namespace
#ifdef LERF_LI_STAMPS
extern "C" void lerf_li_set_stamps(void* p) { g_li_stamps = (unsigned long long*)p; }     // diagnostic build only
#endif

template <int OC, typename TIN, typename TOUT, bool ACC>
static void launch_lut_interp_t(const void* img, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C, int h, int w,
                                Offsets4 off, const int8_t* lut, int interval, void* out, int64_t oy, int64_t ox, int64_t ocs, hipStream_t st) {
    dim3 block(256), grid((w + 15) / 16, (h + 15) / 16, C);
    const float inv_q = 1.0f / (float)(1 << interval);
    if (interval != 4)
        hipLaunchKernelGGL((lut_interp_any_kernel<OC, TIN, TOUT, ACC>), grid, block, 0, st, (const TIN*)img, sy, sx, sc, img_h, img_w, C, h, w, off, lut,
                           interval, (TOUT*)out, oy, ox, ocs, inv_q);
    else
        hipLaunchKernelGGL((lut_interp_kernel<OC, TIN, TOUT, ACC>), grid, block, 0, st, (const TIN*)img, sy, sx, sc, img_h, img_w, C, h, w, off, lut,
                           (TOUT*)out, oy, ox, ocs, inv_q);
}

// the LDS-resident kernel takes the call when it pays (enough positions to amortise a 83.5-KB plane per workgroup) and its
// tile provides for the pattern; returns LERF_EUNSUPPORTED otherwise (the caller falls back to the direct kernels)
static int try_lut_interp_lds(const void* img, int in_dtype, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C,
                              int h, int w, Offsets4 off, const int8_t* lut, int oC, void* out, int out_dtype, int64_t oy, int64_t ox,
                              int64_t ocs, int flags, hipStream_t st) {
    if (flags & LERF_INTERP_DIRECT) return LERF_EUNSUPPORTED;
    if (C > LI_CMAX || (reinterpret_cast<uintptr_t>(lut) & 15)) return LERF_EUNSUPPORTED;
    if (!(flags & LERF_INTERP_LDS) && (int64_t)h * w * C < 65536) return LERF_EUNSUPPORTED;
    int miny = 127, maxy = -127, minx = 127, maxx = -127;
    for (int k = 0; k < 4; ++k) {
        miny = off.dy[k] < miny ? off.dy[k] : miny; maxy = off.dy[k] > maxy ? off.dy[k] : maxy;
        minx = off.dx[k] < minx ? off.dx[k] : minx; maxx = off.dx[k] > maxx ? off.dx[k] : maxx;
    }
    if (maxy - miny > LI_REACH || maxx - minx > LI_REACH) return LERF_EUNSUPPORTED;
    // plane offsets are formed in 32 bits inside a plane
    const int64_t span = (int64_t)(h - 1) * (oy < 0 ? -oy : oy) + (int64_t)(w - 1) * (ox < 0 ? -ox : ox);
    if (span >= (1ll << 30) || sy < 0 || sx < 0 || sc < 0) return LERF_EUNSUPPORTED;
    const int64_t max_off = (int64_t)(C - 1) * sc + (int64_t)(img_h - 1) * sy + (int64_t)(img_w - 1) * sx;
    if (max_off >= (1ll << 31) - 16 || max_off < 3) return LERF_EUNSUPPORTED;
    LiArgs A{};
#ifdef LERF_LI_STAMPS
    A.stamps = g_li_stamps;
#endif
    A.img = img; A.sy = sy; A.sx = sx; A.sc = sc;
    A.max_off = (int64_t)(C - 1) * sc + (int64_t)(img_h - 1) * sy + (int64_t)(img_w - 1) * sx;
    A.img_h = img_h; A.img_w = img_w; A.C = C; A.h = h; A.w = w;
    A.lut = lut; A.out = out; A.ocs = ocs;
    // lanes run along the axis on which the output plane is contiguous
    A.lane_is_y = ((oy == 1 || oy == -1) && !(ox == 1 || ox == -1)) ? 1 : 0;
    A.ol = A.lane_is_y ? oy : ox;
    A.oo = A.lane_is_y ? ox : oy;
    A.miny = miny; A.minx = minx; A.maxx = maxx;
    // 64 x 64 tiles, or 64 x 32 when that leaves a workgroup fewer than four tiles (the pipeline of staging under walking
    // needs a few, and the last round of a short sequence is the whole tail)
    const int cus = li_cu_count();
    const int64_t tiles64 = (int64_t)((h + 63) / 64) * ((w + 63) / 64);
    int TO = tiles64 * oC >= 4ll * cus ? 64 : 32;
    if (flags & LERF_INTERP_TILE64) TO = 64;
    if (flags & LERF_INTERP_TILE32) TO = 32;
    A.tile_h = A.lane_is_y ? LI_TL : TO; A.tile_w = A.lane_is_y ? TO : LI_TL;
    A.th_y = A.tile_h + LI_REACH; A.th_x = A.tile_w + LI_REACH;
    A.tiles_x = (w + A.tile_w - 1) / A.tile_w; A.tiles_y = (h + A.tile_h - 1) / A.tile_h;
    auto odd_pitch = [](int bytes) { int p = (bytes + 3) / 4 * 4; return ((p >> 2) & 1) ? p : p + 4; };
    int rowlen;
    if (C == 1 || (sc == 1 && sx == C)) {                   // rows of (column, channel) bytes, as in memory
        A.layout = (sx == C && (C == 1 || sc == 1)) ? LI_LAYOUT_HWC : LI_LAYOUT_GENERIC;
        rowlen = A.th_x * C; A.rows = A.th_y;
        A.pitch = odd_pitch(rowlen); A.cs = 1; A.xs = C;
    } else {                                                // planes of rows of columns
        A.layout = sx == 1 ? LI_LAYOUT_PLANAR : LI_LAYOUT_GENERIC;
        rowlen = A.th_x; A.rows = A.th_y * C;
        A.pitch = odd_pitch(rowlen); A.cs = A.th_y * A.pitch; A.xs = 1;
    }
    A.groups = (rowlen + 3) / 4;
    A.groups_magic = (unsigned)((1u << 20) / (unsigned)A.groups + 1u);
    A.thy_magic = A.layout == LI_LAYOUT_PLANAR ? (unsigned)((1u << 20) / (unsigned)A.th_y + 1u) : 0u;
    for (int k = 0; k < 4; ++k) A.koff[k] = (off.dy[k] - miny) * A.pitch + (off.dx[k] - minx) * A.xs;
    const int ntiles = A.tiles_x * A.tiles_y;
    int grid = cus;
    if (grid > ntiles * oC) grid = ntiles * oC;
    const bool acc = (flags & LERF_INTERP_ACCUMULATE) != 0;
#define LERF_LL_T(OC, TIN, TOUT, ACC) (TO == 64 ? li_launch<OC, TIN, TOUT, ACC, 64>(A, grid, st) : li_launch<OC, TIN, TOUT, ACC, 32>(A, grid, st))
#define LERF_LL(OC, TIN, TOUT) (acc ? LERF_LL_T(OC, TIN, TOUT, true) : LERF_LL_T(OC, TIN, TOUT, false))
#define LERF_LL_OUT(OC, TIN) (out_dtype == LERF_I16 ? LERF_LL(OC, TIN, int16_t) : out_dtype == LERF_F32 ? LERF_LL(OC, TIN, float) : LERF_LL(OC, TIN, double))
    if (oC == 1) return in_dtype == LERF_U8 ? LERF_LL_OUT(1, uint8_t) : LERF_LL_OUT(1, float);
    return in_dtype == LERF_U8 ? LERF_LL_OUT(3, uint8_t) : LERF_LL_OUT(3, float);
#undef LERF_LL_OUT
#undef LERF_LL
#undef LERF_LL_T
}

// in_dtype: LERF_U8 / LERF_F32; out_dtype: LERF_I16 (numerators) / LERF_F32 / LERF_F64 (values); strides in ELEMENTS, signed
int launch_lut_interp(const void* img, int in_dtype, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C,
                      int h, int w, Offsets4 off, const int8_t* lut, int oC, int interval, void* out, int out_dtype, int64_t oy,
                      int64_t ox, int64_t ocs, int flags, hipStream_t st) {
    if (interval < 1 || interval > 7) return LERF_EUNSUPPORTED;
    if (oC != 1 && oC != 3) return LERF_EUNSUPPORTED;
    if ((in_dtype != LERF_U8 && in_dtype != LERF_F32) || (out_dtype != LERF_I16 && out_dtype != LERF_F32 && out_dtype != LERF_F64))
        return LERF_EUNSUPPORTED;
    if (interval == 4) {
        const int rc = try_lut_interp_lds(img, in_dtype, sy, sx, sc, img_h, img_w, C, h, w, off, lut, oC, out, out_dtype, oy, ox, ocs, flags, st);
        if (rc != LERF_EUNSUPPORTED) return rc;
        if (flags & LERF_INTERP_LDS) return LERF_EUNSUPPORTED;            // the caller insisted on the LDS kernel
    }
    const bool acc = (flags & LERF_INTERP_ACCUMULATE) != 0;
#define LERF_LI(OC, TIN, TOUT)                                                                                                  \
    do {                                                                                                                        \
        if (acc) launch_lut_interp_t<OC, TIN, TOUT, true>(img, sy, sx, sc, img_h, img_w, C, h, w, off, lut, interval, out, oy, ox, ocs, st); \
        else launch_lut_interp_t<OC, TIN, TOUT, false>(img, sy, sx, sc, img_h, img_w, C, h, w, off, lut, interval, out, oy, ox, ocs, st);    \
    } while (0)
#define LERF_LI_OUT(OC, TIN)                                   \
    do {                                                       \
        if (out_dtype == LERF_I16) LERF_LI(OC, TIN, int16_t);  \
        else if (out_dtype == LERF_F32) LERF_LI(OC, TIN, float); \
        else LERF_LI(OC, TIN, double);                         \
    } while (0)
    if (oC == 1) {
        if (in_dtype == LERF_U8) LERF_LI_OUT(1, uint8_t); else LERF_LI_OUT(1, float);
    } else {
        if (in_dtype == LERF_U8) LERF_LI_OUT(3, uint8_t); else LERF_LI_OUT(3, float);
    }
#undef LERF_LI_OUT
#undef LERF_LI
    return LERF_OK;
}

}  // namespace lerf
