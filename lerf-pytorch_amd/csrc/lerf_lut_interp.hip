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
constexpr int LI_NR = 2;                    // wave-rows a compute wave runs interleaved: 2 rows x 2 positions per lane = 20 gathers in flight
constexpr int LI_TL = 128;                  // tile extent along the lane axis: a lane owns two adjacent positions (one 16-byte store)
constexpr int LI_REACH = 3;                 // largest extent of a pattern the LDS tile provides for (s d y: 2, c t: 3)
constexpr int LI_LH = LI_TL + LI_REACH;
constexpr int LI_CMAX = 4;
constexpr int LI_ENTRIES = kL * kL * kL * kL;                 // 83 521
constexpr int LI_LUT_BYTES = (LI_ENTRIES + 63) / 64 * 64;      // 83 584
constexpr int LI_ALL = kStrideA + kStrideB + kStrideC + kStrideD;                   // vertex 4 - vertex 0

// TO = tile extent along the other axis (32; 16: twice the tiles, for launches of few tiles per workgroup)
template <int TO>
struct LiDims {
    static constexpr int OH = TO + LI_REACH;
    static constexpr int ROWS_MAX = LI_CMAX * (LI_LH > OH ? LI_LH : OH);             // staged rows of a planar tile
    // pixel tile: staged rows padded to a multiple of 4 bytes with an odd number of dwords (<= 7 bytes of padding per row)
    static constexpr int PIX_BYTES = (LI_CMAX * LI_LH * OH + 7 * ROWS_MAX + 15) / 16 * 16 + 16;      // + the dump slot
    static constexpr int GROUPS_MAX = LI_CMAX * LI_LH * ((OH + 3) / 4) > LI_CMAX * OH * ((LI_LH + 3) / 4)
                                          ? LI_CMAX * LI_LH * ((OH + 3) / 4) : LI_CMAX * OH * ((LI_LH + 3) / 4);   // 4-pixel groups of a tile
    static constexpr int LDS_BYTES = LI_LUT_BYTES + 2 * PIX_BYTES;
    // tiles whose loads a loader wave keeps in flight: 2 x 11 x 16 bytes per thread for 128 x 16 tiles; a 128 x 32 tile alone is 19 x 16
    static constexpr int DEPTH = TO == 16 ? 2 : 1;
    static_assert(PIX_BYTES < 65536 && LDS_BYTES <= 160 * 1024, "LDS budget");
};

enum { LI_LAYOUT_GENERIC = 0, LI_LAYOUT_HWC = 1, LI_LAYOUT_PLANAR = 2 };

struct LiArgs {
    const void* img; int64_t sy, sx, sc;     // element strides of the image operand [C][img_h][img_w]
    int64_t max_off;                         // largest element offset inside the operand
    int img_h, img_w, C, h, w;
    const int8_t* lut; int lut_planar;       // [17^4][OC] as the reference stores it, or OC planes of LI_LUT_BYTES bytes (LERF_INTERP_LUT_PLANAR)
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
    int stamp_slot;
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

// the two values of a lane (adjacent along the axis on which the plane is contiguous) leave as ONE store: 16 bytes of float64
// per lane, 1 KB per wave-instruction (8-byte stores run at ~7 B/clk/CU, the issue rate of the store path; 16-byte ones twice that)
template <typename TOUT> struct LiPair;
template <> struct LiPair<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct LiPair<float> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct LiPair<int16_t> { typedef short type __attribute__((ext_vector_type(2), aligned(2))); };
template <typename TOUT>
__device__ __forceinline__ TOUT li_value(int acc) {
    if constexpr (sizeof(TOUT) == 2) return (TOUT)acc;
    else return (TOUT)((float)acc * 0.0625f);                     // |acc| <= 2 032: exact in float32
}
// o: the pair's lower address; (lo, hi) in memory order; both = false: the lower only (the frame's last odd position)
template <typename TOUT, bool ACC>
__device__ __forceinline__ void li_store2(TOUT* __restrict__ o, int lo, int hi, bool both) {
    typedef typename LiPair<TOUT>::type P;
    if (both) {
        P v = {li_value<TOUT>(lo), li_value<TOUT>(hi)};
        if constexpr (ACC) v += *reinterpret_cast<P*>(o);
#ifndef LERF_LI_PLAIN_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<P*>(o));
#else
        *reinterpret_cast<P*>(o) = v;
#endif
    } else {
        TOUT v = li_value<TOUT>(lo);
        if constexpr (ACC) v += *o;
        *o = v;
    }
}

// the accumulate form: the pair already in the plane is FETCHED at the top of the pass (li_load2) and added at its end
// (li_store2_pre), so the read's HBM latency runs under the walk instead of in front of the store (a load issued next to the add
// costs the pass its whole latency: the int16 `+=` passes of the call sites ran 54 / 26 us against 45 / 21 for plain stores)
template <typename TOUT>
__device__ __forceinline__ typename LiPair<TOUT>::type li_load2(const TOUT* __restrict__ o, bool both) {
    typedef typename LiPair<TOUT>::type P;
    P v = {(TOUT)0, (TOUT)0};
    if (both) v = *reinterpret_cast<const P*>(o);                 // one exec-masked load, no join of two loads (that would wait)
    return v;                                                     // the frame's last odd position is read where it is stored
}
template <typename TOUT>
__device__ __forceinline__ void li_store2_pre(TOUT* __restrict__ o, int lo, int hi, bool both, typename LiPair<TOUT>::type pre) {
    typedef typename LiPair<TOUT>::type P;
    if (both) {
        P v = {li_value<TOUT>(lo), li_value<TOUT>(hi)};
        v += pre;
#ifndef LERF_LI_PLAIN_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<P*>(o));
#else
        *reinterpret_cast<P*>(o) = v;
#endif
    } else {
        *o = (TOUT)(li_value<TOUT>(lo) + *o);
    }
}

// four pixels -> four bytes, rounded half-to-even and clipped like pixel_value<float>: v_rndne_f32 + v_cvt_pk_u8_f32 (the
// saturating convert with byte insert) per pixel
__device__ __forceinline__ uint32_t li_pack4(const li_v4f& v) {
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(v.x), 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(v.y), 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(v.z), 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(v.w), 3u, r);
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
    // coordinates.  Columns are not: either every column a position of the frame reads exists (columns beyond them hold
    // whatever follows in memory, read only by positions outside the frame), or the tile is of kind 3: the staged columns left
    // and right of the operand are filled in LDS afterwards from the operand's first / last column (li_fix_columns).
    // Kind 2 / 3: the 16-byte groups are loaded no further than the operand's first / last four elements and their bytes
    // shifted back into place (the frame's corners).
    const int xl = T.x0 + A.minx;
    const int x_need = min(T.x0 + A.tile_w, A.w) + A.maxx;                      // one past the last column a valid position reads
    const bool inside = xl >= 0 && x_need <= A.img_w;
    const int nlo = max(0, -xl), hi0 = min(A.th_x, A.img_w - xl);               // staged columns [nlo, hi0) exist in the operand
    T.fast = 0;
    if (A.layout != LI_LAYOUT_GENERIC && (inside || nlo < hi0)) {
        const int yl = min(max(T.y0 + A.miny + A.th_y - 1, 0), A.img_h - 1);      // last staged row, clamped
        const int64_t last = (A.layout == LI_LAYOUT_HWC ? 0 : (int64_t)(A.C - 1) * A.sc) + (int64_t)yl * A.sy + (int64_t)xl * A.sx +
                             (int64_t)A.groups * 4 - 1;
        T.fast = !inside ? 3 : (last <= A.max_off ? 1 : 2);
    }
    return T;
}

// kind-3 tiles: the staged columns outside the operand <- the operand's first / last column, in LDS (edge padding = clamped
// coordinates); by nthr threads, between two barriers
__device__ __forceinline__ void li_fix_columns(const LiArgs& A, const LiTile& T, uint8_t* pix, int lt, int nthr) {
    const int xl = T.x0 + A.minx;
    const int nlo = max(0, -xl), hi0 = min(A.th_x, A.img_w - xl), ncol = nlo + (A.th_x - hi0);
    const int per = A.cs == 1 ? A.C : 1;                                        // HWC tiles: the channels of a pixel lie side by side
    const int n = A.rows * ncol * per;
    for (int i = lt; i < n; i += nthr) {
        const int c = i % per, j = (i / per) % ncol, r = i / (per * ncol);
        const int xx = j < nlo ? j : hi0 + (j - nlo), from = j < nlo ? nlo : hi0 - 1;
        pix[r * A.pitch + xx * A.xs + c] = pix[r * A.pitch + from * A.xs + c];
    }
}

__device__ __forceinline__ void li_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int PENDING>
__device__ __forceinline__ void li_wait_loads_but() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PENDING) : "memory"); }
__device__ __forceinline__ int li_med3(int a, int b, int c) { return min(max(a, b), c); }     // b <= c: clamp (v_med3_i32)

// fast tile -> LDS by NTHR threads (lt = this thread's index among them): every load in flight before the first conversion.
// plan[i] describes the thread's i-th 4-pixel group, the same for every tile (one register per group, built once):
//   bits 0..7 staged row inside its plane (yy), 8..9 plane (c; 0 in HWC tiles), 10..17 group inside the row (g),
//   18..31 dword offset of the group in the LDS tile (threads beyond the tile's last group: the dump slot behind the tile,
//   and the last group's source)
template <typename TIN, int NTHR, int TO>
struct LiStager {
    using D = LiDims<TO>;
    static constexpr int N = (D::GROUPS_MAX + NTHR - 1) / NTHR;
    LiGroup<TIN> pf[N];
    uint32_t plan[N];
    __device__ __forceinline__ void make_plan(const LiArgs& A, int lt) {
        const int ng = A.rows * A.groups;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int q = lt + i * NTHR, qc = min(q, ng - 1);
            const unsigned r = __umul24((unsigned)qc, A.groups_magic) >> 20, g = (unsigned)qc - __umul24(r, (unsigned)A.groups);
            const unsigned c = __umul24(r, A.thy_magic) >> 20, yy = r - __umul24(c, (unsigned)A.th_y);   // HWC: thy_magic = 0
            const unsigned lds = q < ng ? __umul24(r, (unsigned)A.pitch) + g * 4u : (unsigned)(D::PIX_BYTES - 16);
            plan[i] = yy | (c << 8) | (g << 10) | ((lds >> 2) << 18);
        }
    }
    // element offset (32 bits; the host checks the operand's extent and strides) of the group described by p
    template <bool PLANAR>
    __device__ __forceinline__ static int src_raw(const LiArgs& A, uint32_t p, int yl, int xorg) {
        const int gy = li_med3(yl + (int)(p & 255u), 0, A.img_h - 1);               // rows clamp like the pass's coordinates
        int off = __mul24(gy, (int)A.sy) + xorg + (int)((p >> 10) & 255u) * 4;
        if constexpr (PLANAR) off += (int)((p >> 8) & 3u) * (int)A.sc;
        return off;
    }
    template <bool PLANAR>
    __device__ __forceinline__ static int src(const LiArgs& A, uint32_t p, int yl, int xorg, int cap) {
        return li_med3(src_raw<PLANAR>(A, p, yl, xorg), 0, cap);
    }
    template <bool PLANAR>
    __device__ __forceinline__ void issue_t(const LiArgs& A, const LiTile& T) {
        const TIN* __restrict__ img = (const TIN*)A.img;
        const int yl = T.y0 + A.miny, xorg = (T.x0 + A.minx) * (int)A.sx;
        // fast = 2: no group beyond the operand's last four elements; generic tiles (which may start left of the operand): a
        // harmless load.  Every lane loads: an assembly load under a branch leaves its register undefined on the other path,
        // which the compiler answers with a scratch slot and a wait per load.
        const int cap = T.fast == 1 ? 0x7FFFFFFF : (T.fast ? (int)A.max_off - 3 : 0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint32_t p = plan[i];
            asm volatile("" : "+v"(p));        // opaque: or the compiler hoists the fields of all 18 plans out of the tile loop, into scratch
            pf[i].load(img + src<PLANAR>(A, p, yl, xorg, cap));
        }
    }
    __device__ __forceinline__ void issue(const LiArgs& A, const LiTile& T) {
        if (A.cs == 1) issue_t<false>(A, T); else issue_t<true>(A, T);
    }
    // PENDING = loads this wave has issued after this tile's (deeper prefetch: the sets of the following tiles)
    template <int PENDING>
    __device__ __forceinline__ void commit(LiArgs& A, const LiTile& T, uint8_t* pix) {
        li_wait_loads_but<PENDING>();
#ifdef LERF_LI_STAMPS
        if ((threadIdx.x == 0 || threadIdx.x == LI_NCW * 64) && A.stamps && A.stamp_slot < 32) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); A.stamps[(size_t)blockIdx.x * 32 + A.stamp_slot] = t_; }
#endif
        const uint32_t pix_a = li_lds_addr(pix);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            pf[i].landed();
            uint32_t v = pf[i].bytes();
            uint32_t p = plan[i];
            asm volatile("" : "+v"(p));
            if (T.fast >= 2) {                             // wave-uniform: the frame's corners / left and right border
                const int yl = T.y0 + A.miny, xorg = (T.x0 + A.minx) * (int)A.sx;
                const int off = A.cs == 1 ? src_raw<false>(A, p, yl, xorg) : src_raw<true>(A, p, yl, xorg);
                const int sh = off - li_med3(off, 0, (int)A.max_off - 3);          // the load was moved by -sh elements
                // the elements beyond the operand are read by no valid position (kind 2) or filled by li_fix_columns (kind 3)
                v = sh >= 0 ? (sh < 4 ? v >> (8 * sh) : 0u) : (sh > -4 ? v << (8 * -sh) : 0u);
            }
            *(__attribute__((address_space(3))) uint32_t*)(pix_a + ((p >> 18) << 2)) = v;
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
lut_interp_lds_kernel(LiArgs A_) {
    LiArgs A = A_;
    using D = LiDims<TO>;
    extern __shared__ __attribute__((aligned(16))) uint8_t li_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // oC = 3: the three workgroups that walk the same tile sequence (one per plane) read the same pixels at the same time.
    // Workgroups are dealt to the 8 XCDs round-robin (an affinity, nothing depends on it but speed): a triple is made of three
    // workgroups of ONE XCD, so that the tile is fetched from HBM once and served twice by that XCD's L2; the one or two
    // workgroups an XCD has left over form triples across XCDs.
    int plane = 0, slot = (int)blockIdx.x, nslots = (int)gridDim.x;
    if constexpr (OC == 3) {
        const int G = (int)gridDim.x, b = (int)blockIdx.x, xcd = b & 7, j = b >> 3;
        int full_before = 0, left_before = 0, full_all = 0, left_all = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int nx = (G - x + 7) >> 3;
            if (x < xcd) { full_before += nx / 3; left_before += nx % 3; }
            full_all += nx / 3;
            left_all += nx % 3;
        }
        const int nx = (G - xcd + 7) >> 3, full = nx / 3;
        nslots = full_all + left_all / 3;
        if (j < 3 * full) {
            plane = j % 3;
            slot = full_before + j / 3;
        } else {
            const int L = left_before + (j - 3 * full);
            plane = L % 3;
            slot = full_all + L / 3;
            if (slot >= nslots) return;                          // the last one or two workgroups of the launch make no triple
        }
    }
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
        if (OC == 1 || A.lut_planar) {
            const uint4* s = reinterpret_cast<const uint4*>(A.lut + (size_t)plane * LI_LUT_BYTES);
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
        if (tid == 0) li_smem[LI_ENTRIES - 1] = (uint8_t)(OC == 1 || A.lut_planar ? A.lut[(size_t)plane * LI_LUT_BYTES + LI_ENTRIES - 1] : A.lut[(size_t)(LI_ENTRIES - 1) * OC + plane]);
    };
    if (T.fast) {
        LiStager<TIN, LI_NT, TO> S0;
        S0.make_plan(A, tid);
        S0.issue(A, T);
        copy_lut();
        LI_STAMP(1);
        S0.template commit<0>(A, T, pix0);
    } else {
        copy_lut();
        LI_STAMP(1);
        li_stage_generic<TIN>(A, T, pix0, tid, LI_NT);
    }
    __syncthreads();
    if (T.fast == 3) {
        li_fix_columns(A, T, pix0, tid, LI_NT);
        __syncthreads();
    }
    LI_STAMP(2);
    const uint32_t lut_a = li_lds_addr(li_smem);

    // ---- the walks of one tile: wave-rows of 128 positions = (channel, other-axis index), two per lane; LI_NR rows interleaved
    const int lstride = A.lane_is_y ? A.pitch : A.xs, ostride = A.lane_is_y ? A.xs : A.pitch;
    const int nl = A.lane_is_y ? A.h : A.w, no = A.lane_is_y ? A.w : A.h;
    TOUT* const outp = (TOUT*)A.out;
    const int nrows = A.C * TO;
    const bool rev = A.ol < 0;                                             // the plane runs backwards along the lane axis (rot 2, 3)
    auto out_row = [&](int c, int o) { return outp + ((int64_t)(c * OC + plane) * A.ocs + (int64_t)o * A.oo); };
    auto compute = [&](const LiTile& Tc, const uint8_t* pix) {
        const int l0 = A.lane_is_y ? Tc.y0 : Tc.x0, o0 = A.lane_is_y ? Tc.x0 : Tc.y0;
        const int lpos = l0 + 2 * lane;                                    // this lane's positions: lpos, lpos + 1
        const bool ok0 = lpos < nl, ok1 = lpos + 1 < nl;
        // lower address of the pair: position lpos when the plane runs forwards, lpos + 1 when backwards
        const int lane_out = (int)((int64_t)(rev && ok1 ? lpos + 1 : lpos) * A.ol);
        const uint32_t pa0 = li_lds_addr(pix) + (uint32_t)(2 * lane * lstride), pa1 = pa0 + (uint32_t)lstride;
        const int ovalid = min(TO, no - o0);                               // rows of the tile inside the frame
        auto one_row = [&](int c, int o) {
            const uint32_t s0 = (uint32_t)(c * A.cs + o * ostride);
            uint32_t p[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { p[0][k] = li_pixel_hi(pa0 + (s0 + A.koff[k])); p[1][k] = li_pixel_hi(pa1 + (s0 + A.koff[k])); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]), "+v"(p[1][0]), "+v"(p[1][1]), "+v"(p[1][2]), "+v"(p[1][3]));
            const LiWalk W0 = li_walk(lut_a, p[0][0], p[0][1], p[0][2], p[0][3]);
            const LiEntries E0 = li_gather(W0);
            const LiWalk W1 = li_walk(lut_a, p[1][0], p[1][1], p[1][2], p[1][3]);
            const LiEntries E1 = li_gather(W1);
            const int n0 = li_numerator(W0, E0), n1 = li_numerator(W1, E1);
#ifdef LERF_LI_NO_STORE
            if (n0 == 0x7FFFFFF && ok0)
#else
            if (ok0)
#endif
                li_store2<TOUT, ACC>(out_row(c, o0 + o) + lane_out, rev && ok1 ? n1 : n0, rev ? n0 : n1, ok1);
        };
#pragma unroll 1
        for (int r0 = wave; r0 < nrows; r0 += LI_NCW * LI_NR) {
            int c[LI_NR], o[LI_NR];
            bool all = true;
#pragma unroll
            for (int j = 0; j < LI_NR; ++j) {
                const int r = r0 + j * LI_NCW;
                c[j] = r / TO;
                o[j] = r - c[j] * TO;
                all = all && r < nrows && o[j] < ovalid;                   // wave-uniform
            }
            if (all) {
                TOUT* op[LI_NR];
                [[maybe_unused]] typename LiPair<TOUT>::type pre[LI_NR];
#pragma unroll
                for (int j = 0; j < LI_NR; ++j) op[j] = out_row(c[j], o0 + o[j]) + lane_out;
                if constexpr (ACC) {
#pragma unroll
                    for (int j = 0; j < LI_NR; ++j) pre[j] = li_load2<TOUT>(op[j], ok1);
                    __builtin_amdgcn_sched_barrier(0);                      // issued here, consumed after the walks
                }
                uint32_t px[LI_NR][2][4];
#pragma unroll
                for (int j = 0; j < LI_NR; ++j) {
                    const uint32_t s0 = (uint32_t)(c[j] * A.cs + o[j] * ostride);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t sk = s0 + (uint32_t)A.koff[k];      // scalar
                        px[j][0][k] = li_pixel_hi(pa0 + sk);
                        px[j][1][k] = li_pixel_hi(pa1 + sk);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(px[0][0][0]), "+v"(px[0][0][1]), "+v"(px[0][0][2]), "+v"(px[0][0][3]), "+v"(px[0][1][0]), "+v"(px[0][1][1]), "+v"(px[0][1][2]), "+v"(px[0][1][3]),
                               "+v"(px[1][0][0]), "+v"(px[1][0][1]), "+v"(px[1][0][2]), "+v"(px[1][0][3]), "+v"(px[1][1][0]), "+v"(px[1][1][1]), "+v"(px[1][1][2]), "+v"(px[1][1][3]));
                LiWalk W[LI_NR][2];
                LiEntries E[LI_NR][2];
#pragma unroll
                for (int j = 0; j < LI_NR; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
#ifdef LERF_LI_NO_WALK
                        W[j][q].k0 = px[j][q][0]; W[j][q].k1 = px[j][q][1]; W[j][q].k2 = px[j][q][2]; W[j][q].k3 = px[j][q][3];
                        E[j][q].e0 = px[j][q][0]; E[j][q].e1 = px[j][q][1]; E[j][q].e2 = px[j][q][2]; E[j][q].e3 = px[j][q][3]; E[j][q].e4 = px[j][q][0];
#else
                        W[j][q] = li_walk(lut_a, px[j][q][0], px[j][q][1], px[j][q][2], px[j][q][3]);
                        E[j][q] = li_gather(W[j][q]);
                        __builtin_amdgcn_sched_barrier(0);                  // this walk's gathers fly under the next walk
#endif
                    }
#pragma unroll
                for (int j = 0; j < LI_NR; ++j) {
                    const int n0 = li_numerator(W[j][0], E[j][0]), n1 = li_numerator(W[j][1], E[j][1]);
#ifdef LERF_LI_NO_STORE                                                     // ablation builds (tools/build_li_variant.sh): never the product
                    if (n0 == 0x7FFFFFF && ok0)
#else
                    if (ok0)
#endif
                    {
                        if constexpr (ACC) li_store2_pre<TOUT>(op[j], rev && ok1 ? n1 : n0, rev ? n0 : n1, ok1, pre[j]);
                        else li_store2<TOUT, false>(op[j], rev && ok1 ? n1 : n0, rev ? n0 : n1, ok1);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < LI_NR; ++j) {
                    const int r = r0 + j * LI_NCW;
                    if (r < nrows && o[j] < ovalid) one_row(c[j], o[j]);
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
#ifndef LERF_LI_NO_PRIO
        __builtin_amdgcn_s_setprio(3);                                  // one loader wave beside three walking waves per SIMD: it goes first
#endif
        const int lt = tid - LI_NCW * 64;
        using Stager = LiStager<TIN, LI_NLW * 64, TO>;
        Stager S[D::DEPTH];
        LiTile Ts[D::DEPTH];
        int cur = 0;
        S[0].make_plan(A, lt);
#pragma unroll
        for (int d = 1; d < D::DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < Stager::N; ++i) S[d].plan[i] = S[0].plan[i];
#pragma unroll
        for (int d = 0; d < D::DEPTH; ++d) {
            const int td = t + (1 + d) * nslots;
            Ts[d] = li_tile_of(A, td < ntiles ? td : t);
            S[d].issue(A, Ts[d]);                                       // (a tile of the generic kind loads nothing useful)
        }
        bool done = false, fix_now = false;         // (the first tile's columns were filled at start-up)
#pragma unroll 1
        while (!done) {
#pragma unroll
            for (int d = 0; d < D::DEPTH; ++d) {
                if (t + nslots >= ntiles) { done = true; break; }
                if (fix_now) __syncthreads();                               // the walking waves fill this period's tile's border columns
                uint8_t* nxt = pix0 + (cur ^ 1) * D::PIX_BYTES;
#ifdef LERF_LI_STAMPS
                A.stamp_slot = 14 + 3 * ((stamp_k - 3) / 2);
#endif
                if (Ts[d].fast) S[d].template commit<(D::DEPTH - 1) * Stager::N>(A, Ts[d], nxt);
                else { li_wait_loads(); li_stage_generic<TIN>(A, Ts[d], nxt, lt, LI_NLW * 64); }
                LI_STAMP_L(15 + 3 * ((stamp_k - 3) / 2));
                const int tnn = t + (1 + D::DEPTH) * nslots;
                Ts[d] = li_tile_of(A, tnn < ntiles ? tnn : t);
                S[d].issue(A, Ts[d]);
                LI_STAMP_L(16 + 3 * ((stamp_k - 3) / 2));
                LI_STAMP_L(stamp_k + 1);
                stamp_k += 2;
                __syncthreads();
                cur ^= 1;
                t += nslots;
                fix_now = li_tile_of(A, t).fast == 3;
            }
        }
        if (fix_now) __syncthreads();
        li_wait_loads();                                                // nothing of this wave may be in flight at its end
#else
        bool fix_now = false;
        while (t + nslots < ntiles) { if (fix_now) __syncthreads(); __syncthreads(); t += nslots; fix_now = li_tile_of(A, t).fast == 3; }
        if (fix_now) __syncthreads();
#endif
    } else {
        int cur = 0;
        bool first = true;
#pragma unroll 1
        for (;;) {
            if (!first && T.fast == 3) {
                li_fix_columns(A, T, pix0 + cur * D::PIX_BYTES, tid, LI_NCW * 64);
                __syncthreads();
            }
            first = false;
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
    if (max_off >= (1ll << 31) - 16 || max_off < 3 || sy >= (1 << 23) || img_h >= (1 << 23)) return LERF_EUNSUPPORTED;
    LiArgs A{};
#ifdef LERF_LI_STAMPS
    A.stamps = g_li_stamps;
#endif
    A.img = img; A.sy = sy; A.sx = sx; A.sc = sc;
    A.max_off = (int64_t)(C - 1) * sc + (int64_t)(img_h - 1) * sy + (int64_t)(img_w - 1) * sx;
    A.img_h = img_h; A.img_w = img_w; A.C = C; A.h = h; A.w = w;
    A.lut = lut; A.lut_planar = (flags & LERF_INTERP_LUT_PLANAR) ? 1 : 0; A.out = out; A.ocs = ocs;
    // lanes run along the axis on which the output plane is contiguous
    A.lane_is_y = ((oy == 1 || oy == -1) && !(ox == 1 || ox == -1)) ? 1 : 0;
    A.ol = A.lane_is_y ? oy : ox;
    A.oo = A.lane_is_y ? ox : oy;
    A.miny = miny; A.minx = minx; A.maxx = maxx;
    // 128 x 32 tiles, or 128 x 16 when that leaves a workgroup fewer than four tiles (the pipeline of staging under walking
    // needs a few, and the last round of a short sequence is the whole tail)
    const int cus = li_cu_count();
    const int nl_ = A.lane_is_y ? h : w, no_ = A.lane_is_y ? w : h;
    const int64_t tiles32 = (int64_t)((nl_ + LI_TL - 1) / LI_TL) * ((no_ + 31) / 32);
    int TO = tiles32 * oC >= 4ll * cus ? 32 : 16;
    if (flags & LERF_INTERP_TILE64) TO = 32;
    if (flags & LERF_INTERP_TILE32) TO = 16;
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
#define LERF_LL_T(OC, TIN, TOUT, ACC) (TO == 32 ? li_launch<OC, TIN, TOUT, ACC, 32>(A, grid, st) : li_launch<OC, TIN, TOUT, ACC, 16>(A, grid, st))
#define LERF_LL(OC, TIN, TOUT) (acc ? LERF_LL_T(OC, TIN, TOUT, true) : LERF_LL_T(OC, TIN, TOUT, false))
#define LERF_LL_OUT(OC, TIN) (out_dtype == LERF_I16 ? LERF_LL(OC, TIN, int16_t) : out_dtype == LERF_F32 ? LERF_LL(OC, TIN, float) : LERF_LL(OC, TIN, double))
    if (oC == 1) return in_dtype == LERF_U8 ? LERF_LL_OUT(1, uint8_t) : LERF_LL_OUT(1, float);
    return in_dtype == LERF_U8 ? LERF_LL_OUT(3, uint8_t) : LERF_LL_OUT(3, float);
#undef LERF_LL_OUT
#undef LERF_LL
#undef LERF_LL_T
}

// ---------------------------------------------------------------------------
// the call sites' epilogue of a LUT stage on the summed NUMERATORS (resample/eval_lut_sr.py:573-577, 621-628):
//   np.round(np.clip(pred / avg_factor + bias, 0, norm)).astype(np.float32)
// pred = numerators / q (exact in float64); every step is the float64 operation numpy performs, in numpy's order, no
// contraction -- the float32 array the caller gets is bit for bit numpy's.  The steps come as a small program.
// ---------------------------------------------------------------------------
struct EpiProgram { int n; int op[LERF_EPI_MAX_OPS]; double a[LERF_EPI_MAX_OPS], b[LERF_EPI_MAX_OPS]; };

__global__ void __launch_bounds__(256)
numer_epilogue_kernel(const int16_t* __restrict__ acc, int64_t n, double inv_q, EpiProgram P, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    int16_t v[4] = {0, 0, 0, 0};
    const bool full = i0 + 4 <= n && (reinterpret_cast<uintptr_t>(acc) & 7) == 0;
    if (full) {
        const uint2 w = *reinterpret_cast<const uint2*>(acc + i0);
        v[0] = (int16_t)(w.x & 0xFFFFu); v[1] = (int16_t)(w.x >> 16); v[2] = (int16_t)(w.y & 0xFFFFu); v[3] = (int16_t)(w.y >> 16);
    } else {
        for (int k = 0; k < 4; ++k) if (i0 + k < n) v[k] = acc[i0 + k];
    }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = (double)v[k] * inv_q;                       // exact: q is a power of two
        for (int j = 0; j < P.n; ++j) {
            switch (P.op[j]) {
                case LERF_EPI_DIV: x = x / P.a[j]; break;
                case LERF_EPI_MUL: x = x * P.a[j]; break;
                case LERF_EPI_ADD: x = x + P.a[j]; break;
                case LERF_EPI_CLIP: x = fmin(fmax(x, P.a[j]), P.b[j]); break;
                default: x = __builtin_rint(x); break;          // LERF_EPI_ROUND: half to even, np.round
            }
        }
        r[k] = (float)x;                                       // astype(np.float32): round to nearest even
    }
    if (full && (reinterpret_cast<uintptr_t>(out) & 15) == 0) *reinterpret_cast<float4*>(out + i0) = make_float4(r[0], r[1], r[2], r[3]);
    else for (int k = 0; k < 4; ++k) if (i0 + k < n) out[i0 + k] = r[k];
}

int launch_numer_epilogue(const int16_t* acc, int64_t n, int interval, const lerf_epi_op_t* ops, int n_ops, float* out, hipStream_t st) {
    if (n_ops < 0 || n_ops > LERF_EPI_MAX_OPS || interval < 1 || interval > 7) return LERF_EINVAL;
    EpiProgram P{};
    P.n = n_ops;
    for (int j = 0; j < n_ops; ++j) {
        if (ops[j].op < LERF_EPI_DIV || ops[j].op > LERF_EPI_ROUND) return LERF_EINVAL;
        P.op[j] = ops[j].op; P.a[j] = ops[j].a; P.b[j] = ops[j].b;
    }
    const int64_t blocks = (n + 1023) / 1024;
    if (blocks > 0x7FFFFFFF) return LERF_EUNSUPPORTED;
    hipLaunchKernelGGL(numer_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, st, acc, n, 1.0 / (double)(1 << interval), P, out);
    return LERF_OK;
}

// in_dtype: LERF_U8 / LERF_F32; out_dtype: LERF_I16 (numerators) / LERF_F32 / LERF_F64 (values); strides in ELEMENTS, signed
int launch_lut_interp(const void* img, int in_dtype, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C,
                      int h, int w, Offsets4 off, const int8_t* lut, int oC, int interval, void* out, int out_dtype, int64_t oy,
                      int64_t ox, int64_t ocs, int flags, hipStream_t st) {
    if (interval < 1 || interval > 7) return LERF_EUNSUPPORTED;
    if (oC != 1 && oC != 3) return LERF_EUNSUPPORTED;
    if ((in_dtype != LERF_U8 && in_dtype != LERF_F32) || (out_dtype != LERF_I16 && out_dtype != LERF_F32 && out_dtype != LERF_F64))
        return LERF_EUNSUPPORTED;
    if ((flags & LERF_INTERP_LUT_PLANAR) && (interval != 4 || (flags & LERF_INTERP_DIRECT))) return LERF_EINVAL;
    if (interval == 4) {
        const int rc = try_lut_interp_lds(img, in_dtype, sy, sx, sc, img_h, img_w, C, h, w, off, lut, oC, out, out_dtype, oy, ox, ocs, flags, st);
        if (rc != LERF_EUNSUPPORTED) return rc;
        if (flags & (LERF_INTERP_LDS | LERF_INTERP_LUT_PLANAR)) return LERF_EUNSUPPORTED;   // the caller insisted on the LDS kernel (a planar LUT is its format)
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
