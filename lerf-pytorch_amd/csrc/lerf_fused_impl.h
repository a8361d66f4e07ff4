// Implementation of the tile-fused kernels, included once per channel count: the including translation unit defines
//   LERF_FUSED_NS   namespace of this instance (fused = RGB, fused_c1 = single channel, fused_c4 = four channels)
//   LERF_FUSED_CH   channels per pixel; the LR tile is 64 rows x (192 / CH) pixels, 192 pixel-channels per tile row
//   LERF_FUSED_TH   tile rows (default 64)
// (a plain C++ namespace-parametrised include: everything in here is gfx950 code, there is no second platform).
//
// Tile-fused uint8 SR path for MI355X (gfx950): stage-1 LUTs -> stage-2 LUTs -> spatially-varying resampling per
// 64x64 LR tile (tiles handed to the XCDs in contiguous eighths: xcd_order).  With a caller workspace (the normal case)
// it is TWO launches: s1_kernel computes stage 1 once per pixel and parks its uint8 output in the workspace,
// sr_fused_kernel<.., FROM_FEAT> runs stage 2, the finalisation and stage 3 from it -- the hyper-parameters never leave
// the CU.  Without a workspace ONE launch does everything and recomputes stage 1 on each tile's halo.
//
// Reference path being replaced: eltr._worker, resample/eval_lut_sr.py:541-665
// (FourSimplexInterpFaster :24-470 x 24 passes, SteeringGaussianResize2dNumpy /
// AmplifiedLinearResize2dNumpy, resize_right/resize_right2d_numpy.py:142-282).
//
// Work decomposition
//   one 1024-thread workgroup (16 waves, 1 per CU: LDS-bound) per 64x64 LR tile
//   of one frame.  The tile owns the output pixels whose support starts inside
//   it.  Halo: 3 (stage 1) + 3 (stage 2) + S/2 (stage 3) LR pixels per side,
//   recomputed per tile; true image borders use the reference's rules (clamped
//   sampling for the LUT stages, zero image / edge hyper for stage 3).
//
// LDS plan (dynamic, one array; sizes for S=2 / S=4)
//   B    feat tile u8                   (72x72x3 = 15.5 KB / 74x74x3)     all stages
//   LUT  stage 1: one int8 LUT          83.5 KB
//        stage 2 (LeRF-G): a THIRD of one packed LUT = the 7 top-axis levels that pixels whose
//        centre level lies in the bin can touch, 7*4913 dwords = 134 KB
//   C    input tile u8 (78x78x3)        stage 1 only
//   ACC  int16 partial sums             stage 1 (and stage 2 of LeRF-L)
//   LST  pixel lists sorted by bin      stage 2 (LeRF-G), transient (under the piece, which waits in registers)
//   D    (hq0,hq1,hq2,feat) dwords      stage 3, overlays LUT (52 KB / 55 KB)
//   GEO  tile geometry tables           stage 3 (9 KB; staged at kernel start for S=2)
//   TQ   rounding-tie queue             stage 3 (8 KB, behind D)
//   OUT  the tile's output block        stage 3, block tasks (2 x 2 outputs that share their taps: integer x2 LeRF-G, LeRF-L):
//        assembled byte by byte (132 x 416 B behind TQ), ties patched in it, stored as whole 16-byte chunks
//   CGRP column-group table             stage 3, LeRF-L block tasks (behind OUT)
//
// Stage 2 of LeRF-G keeps the whole 3-channel LUT entry in one dword, so one
// simplex walk (index sort + 5 LDS gathers) serves all three hyper channels.
// The full packed LUT (326 KB) cannot live in LDS; pixels are therefore binned
// by the level of their centre value (which selects the slowest LUT axis
// for every mode and rotation), and each (LUT, bin) phase processes only the
// pixels of that bin with all 64 lanes busy.  Per-pixel accumulators stay in
// VGPRs across phases (two packed 16-bit fields + one).
#include <string.h>
#include <atomic>
#include <type_traits>

#include "lerf_kernels.h"
#include "lerf_stage3.h"
#include "lerf_warp_px.h"

namespace lerf {
namespace LERF_FUSED_NS {

#ifndef LERF_FUSED_NT
#define LERF_FUSED_NT 1024         // threads per workgroup
#endif
constexpr int NT = LERF_FUSED_NT;  // threads per workgroup
constexpr int NW = NT / 64;        // waves
constexpr int CH = LERF_FUSED_CH;  // channels per pixel of this instance
static_assert(CH == 1 || CH == 3 || CH == 4, "192 pixel-channels per tile row");
#ifndef LERF_FUSED_TH
#define LERF_FUSED_TH 64           // tile rows: 64 = the throughput tile; the 32- and 16-row instances (lerf_fused_h32 / _h16.hip) serve launches
#endif                             // too small to fill the chip with 64-row tiles (a 256 x 256 frame: 16 / 32 / 64 workgroups)
constexpr int TH = LERF_FUSED_TH, TW = 192 / CH;   // LR tile: TH rows x 192 pixel-channels (64x64 RGB pixels, 64x192 grey, 64x48 RGBA):
                                   // the same number of stage-2 positions, registers and LDS bytes for every channel count
constexpr int R1 = 3, R2 = 3;      // stage radii: the reach of the widest sampling patterns (c, t: 3 px; s 1, d 2, y 2)
constexpr int MAXM = 4;            // modes per stage the 16-bit sums can hold (4 rotations x 4 modes x 16 x 255 < 2^16)
constexpr int LUT_PAD = 83584;     // padded entries per LUT in the pack (16-B multiple)
// Stage-2 bins: a pixel-channel goes by the top-axis level (high nibble) of its centre value, which selects the slowest
// LUT axis for every mode and rotation.  Bin b covers the levels [bin_lo(b), bin_lo(b + 1)); its piece of a packed LUT is
// those levels plus the one above (the simplex walk steps up once).
#ifndef LERF_NBIN
#define LERF_NBIN 3
#endif
constexpr int NBIN = LERF_NBIN;
#if LERF_NBIN == 4      // four bins of 4 levels, 5-level pieces of 96 KiB (24 phases per tile): the measured alternative
__host__ __device__ constexpr int bin_lo(int b) { return 4 * b; }
__device__ __forceinline__ uint32_t bin_of_level(uint32_t msb) { return msb >> 2; }
constexpr int PIECE_LEVELS = 5;                               // levels per piece
#else                   // three bins of 6, 5 and 5 levels, 7-level pieces of 134 KiB that fill the LDS: 18 longer phases
__host__ __device__ constexpr int bin_lo(int b) { return b == 0 ? 0 : (b == 1 ? 6 : (b == 2 ? 11 : 16)); }
__device__ __forceinline__ uint32_t bin_of_level(uint32_t msb) { return (msb >= 6u ? 1u : 0u) + (msb >= 11u ? 1u : 0u); }
constexpr int PIECE_LEVELS = 7;
#endif
constexpr int PIECE_ENTRIES = PIECE_LEVELS * kStrideA;
constexpr int NSLAB = (PIECE_ENTRIES * 4 + 16 * NT - 1) / (16 * NT);   // 16 bytes per thread and slab
constexpr int PIECE_BYTES = NSLAB * 16 * NT;                  // one piece in the pack (144 KiB, 34391 dwords used)
constexpr int PIECE_BLOCKS = (PIECE_ENTRIES * 4 + 1023) / 1024;      // 1-KiB blocks of a piece that hold data
constexpr int PIECE_LDS = PIECE_BLOCKS * 1024;                // what a piece occupies in LDS

// BIG = false: geometry tables for scale factors up to 4.9, staged at kernel start for the 2x2 support of RGB frames (the
// headline layout).  BIG = true: up to x8.1, always staged late over the dead LUT area.
// NOHALO = true (the EMIT kernels): the hyper region is the tile itself -- the stage-3 ring of R3 pixels is looked up only by
// the kernels that run stage 3 (the packed maps a warp reads hold every pixel once); the feat region keeps its size, so the
// aligned-dword tile load stays.
template <int S, bool BIG = false, bool NOHALO = false>
struct Dims {
    static constexpr int R3 = S / 2;
    static constexpr int HR = NOHALO ? 0 : R3;                                             // ring of the hyper region around the tile
    static constexpr int HY = TH + 2 * HR, HX = TW + 2 * HR, HP = HX * CH, NH = HY * HP;   // hyper region
    static constexpr int FY = TH + 2 * R3 + 2 * R2, FX = TW + 2 * R3 + 2 * R2, FP = FX * CH, NF = FY * FP;   // feat region
    static constexpr int HO = R2 + R3 - HR;                                                // the hyper region's origin inside the feat region
    // input region; its LDS pitch is a dword multiple with room for a 0..3 byte phase, so that interior tiles can be
    // fetched as aligned dwords (row r of the tile = global bytes from the 4-byte boundary below its first pixel)
    static constexpr int IY = FY + 2 * R1, IX = FX + 2 * R1, IPB = IX * CH, IP = (IPB + 3 + 3) / 4 * 4, NI = IY * IP;
    static constexpr int up16(int x) { return (x + 15) / 16 * 16; }
    static constexpr int OFF_B = 0;
    static constexpr int OFF_X = up16(NF);                       // stage-dependent area starts here
    // stage 1 (and byte-LUT stage 2)
    static constexpr int OFF_LUT = OFF_X;
    static constexpr int OFF_C = OFF_LUT + LUT_PAD;
    static constexpr int OFF_ACC = OFF_C + up16(NI);
    static constexpr int END1 = OFF_ACC + up16(NF * 2);
    // stage 2, LeRF-G: one piece.  The position lists and the wave x bin count table only live between the binning and the
    // first piece store (the first piece waits in registers) and overlay the piece.
    // positions that are looked up: the hyper region without its last row and column (never read by an owned output, see the binning)
#ifndef LERF_FULL_RING
    static constexpr int NHL = HR > 0 ? (HY - 1) * (HX - 1) * CH : NH;
#else
    static constexpr int NHL = NH;
#endif
    static constexpr int MAXR = (NHL + NBIN * 63 + NT - 1) / NT;  // slot rounds: every bin padded to whole waves (13 for S = 2, 14 for S = 4)
    static constexpr int OFF_LST = OFF_X;
    static constexpr int OFF_TAB = OFF_LST + MAXR * NT * 2;       // [wave][4 bins] counts
    static_assert(OFF_TAB + NW * 4 * 4 <= OFF_X + PIECE_LDS, "binning scratch fits under the piece");
    static constexpr int END2 = OFF_X + PIECE_LDS;
    // stage 3
    static constexpr int OFF_D = OFF_X;
    // S = 2, RGB, scale <= 4.9: the tile geometry is staged at kernel start (its table look-ups overlap the input load) into
    // a region of its own; the other layouts have no room for that and stage it right before stage 3, over the dead LUT area
    static constexpr bool GEO_EARLY = (S == 2) && CH == 3 && !BIG;
    static constexpr int cmax0(int a, int b) { return a > b ? a : b; }
    static constexpr int OFF_GEO = GEO_EARLY ? cmax0(END1, END2) : OFF_D + up16(NH * 4);
    static constexpr int GEO_ROWS = BIG ? TH * 8 + 8 : TH * 5;    // max owned output rows per tile
    static constexpr int GEO_COLS = BIG ? TW * 8 + 8 : TW * 5;    // max owned output columns per tile
    static constexpr int SZ_GEO = (GEO_ROWS + GEO_COLS) * (4 + 4 * S) + (GEO_ROWS + 16) * 4;   // + row-group table
    static constexpr int END3 = OFF_GEO + SZ_GEO;
    // stage 3: queue of the outputs that sit on a rounding tie (re-evaluated in float64 after the task loop, all lanes
    // busy, instead of one lane at a time inside it); over the dead LUT piece, behind D (and behind the late geometry)
    static constexpr int TQ_CAP = 2048;                           // (lerf_sr_geo_t.tie_queue_cap lowers the part in use: tests)
    static constexpr int OFF_TQ = GEO_EARLY ? OFF_D + up16(NH * 4) : END3;
    static_assert(OFF_TQ + TQ_CAP * 4 <= OFF_X + PIECE_LDS, "tie queue fits under the piece");
    static_assert(GEO_EARLY || END3 <= OFF_X + PIECE_LDS, "late geometry fits under the piece");
    // stage 3, block tasks (integer x2 tiles: 2 x 2 outputs share their taps): the tile's output bytes are assembled in LDS
    // (behind the tie queue, over the dead LUT piece) and leave as whole 16-byte chunks; the rows sit at the phase of their
    // global address modulo 16
    static constexpr int OUT_ROWS = 2 * TH + 4, OUT_PITCH = up16(2 * TW * CH + 16 + 15);
    static constexpr int OFF_OUT = up16(OFF_TQ + TQ_CAP * 4);
    static constexpr int OFF_CGRP = OFF_OUT + OUT_ROWS * OUT_PITCH;          // column-group table of the LeRF-L block tasks
    static constexpr bool OUT_FITS = OFF_CGRP + (2 * TW + 16) * 4 <= OFF_X + PIECE_LDS;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int LDS_BYTES = cmax(END1, cmax(END2, GEO_EARLY ? END3 : 0)) + 512;  // + small control block
    static constexpr int OFF_CTL = LDS_BYTES - 512;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KiB of LDS");
    static_assert(NT != 1024 || (MAXR <= 14 && (NH + NT - 1) / NT <= 15), "slot rounds / positions per thread the register budget was sized for");
};

// Frames of DIFFERENT sizes in one launch (general kernels only): the descriptors travel in the kernel-argument segment, so
// a workgroup finds its frame with scalar loads.  first_block = index of the frame's first tile in the launch.
constexpr int RAGGED_MAX = 16;
struct FrameDesc {
    const uint8_t* img; uint8_t* out; uint8_t* feat; uint32_t* emit;
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;
    int H, W, oH, oW, tiles_x, tiles_y, first_block, pad_;
};
struct RaggedTable { int n; int pad_; FrameDesc d[RAGGED_MAX]; };
struct NoTable {};

struct Params {
    const uint8_t* img; int64_t in_sn;
    uint8_t* out; int64_t out_sn;
    int H, W, oH, oW, tiles_y, tiles_x;
    int ty_org, tx_org;              // LR origin of tile (0, 0): the region of interest of lerf_sr_geo_t (0, 0 = whole frame)
    const uint8_t* pack;             // fused LUT pack (see lerf_fused_lutpack_*)
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;      // tie guard (may be NULL: guard off)
    float max_sigma;
    int n1, n2;                      // modes per stage (1..MAXM); the pack holds n1 stage-1 and 2 * n2 stage-2 LUTs
    int s2off[2 * MAXM][6];          // stage-2 LUT l: feat-tile byte offsets of pixels b,c,d for rotations par, par+2
    int s1off[MAXM][12];             // general kernels: stage-1 LUT m: input-tile byte offsets of b,c,d for rotations 0..3
    int s1dyx[MAXM][12];             // the same pixels as (dy << 16) | (dx & 0xFFFF) in the frame (vector-memory pixel path)
    int s2dyx[2 * MAXM][6];          // stage-2 LUT l, rotations par / par + 2
    uint32_t* emit; int64_t emit_sn;   // EMIT kernels: packed stage outputs, frame stride in dwords
    uint8_t* feat; int64_t feat_sn;    // two-launch path: stage-1 output [N][H][W][C] between s1_kernel and the FROM_FEAT kernel
    unsigned long long* stamps;      // diagnostic builds (-DLERF_STAMPS) only: [blocks][16] cycle stamps
    int tq_cap;                      // stage-3 tie queue entries in use (<= Dims::TQ_CAP; lerf_sr_geo_t.tie_queue_cap)
    int pad_mode;                    // LERF_PAD_* of the image operand of stage 3 at the true frame borders
    int host_input;                  // the input frames live in (pinned) HOST memory: read every pixel once (LDS tile), never 39 times
    int out_pitch;                   // bytes per output row (lerf_sr_geo_t.out_row_pitch; 0 = dense rows of oW * CH bytes)
    WarpGeo wgeo; const int32_t* wboxes;   // tile-fused warp (LERF_FUSED_WARP): homography + per-tile output boxes {i0, i1, j0, j1}
};

#ifdef LERF_STAMPS
#define LERF_STAMP(k) do { if (tid == 0) P.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define LERF_STAMP_ADD(k, t0) do { if (tid == 0) P.stamps[(size_t)blockIdx.x * 16 + (k)] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#define LERF_NOW() __builtin_amdgcn_s_memtime()
// the constant 100 MHz counter beside the shader-clock one: (s_memtime ticks) / (s_memrealtime ticks) x 100 MHz = the clock
// the chip ran this tile at (tools/stamps.py prints it; slots 13 / 14 of a tile's stamps)
#define LERF_STAMP_RT(k) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); \
                                              P.stamps[(size_t)blockIdx.x * 16 + (k)] = t_; } } while (0)
#else
#define LERF_STAMP_RT(k) do {} while (0)
#define LERF_STAMP(k) do {} while (0)
#define LERF_STAMP_ADD(k, t0) do {} while (0)
#define LERF_NOW() 0ull
#endif

// byte offsets of the 3 non-centre pixels of (mode, rot) in a u8 tile of pitch P
struct Off3 { int o[3]; };
template <int P>
__host__ __device__ constexpr Off3 tile_offsets(char mode, int rot) {
    int dy[4] = {0, 0, 0, 0}, dx[4] = {0, 0, 0, 0};
    switch (mode) {                                                  // resample/eval_lut_sr.py:30-81
        case 's': dy[1] = 0; dx[1] = 1; dy[2] = 1; dx[2] = 0; dy[3] = 1; dx[3] = 1; break;
        case 'd': dy[1] = 0; dx[1] = 2; dy[2] = 2; dx[2] = 0; dy[3] = 2; dx[3] = 2; break;
        case 'y': dy[1] = 1; dx[1] = 1; dy[2] = 1; dx[2] = 2; dy[3] = 2; dx[3] = 1; break;
        case 'c': dy[1] = 0; dx[1] = 1; dy[2] = 0; dx[2] = 2; dy[3] = 0; dx[3] = 3; break;
        default:  dy[1] = 1; dx[1] = 1; dy[2] = 2; dx[2] = 2; dy[3] = 3; dx[3] = 3; break;   // 't'
    }
    Off3 r{};
    for (int k = 1; k < 4; ++k) {
        int y = dy[k], x = dx[k];
        for (int i = 0; i < (rot & 3); ++i) { int t = y; y = x; x = -t; }
        r.o[k - 1] = y * P + x * CH;
    }
    return r;
}

// ---------------------------------------------------------------------------
// simplex walk (index/weight computation only).  Keys are (LSB << 16) | axis stride so that one unsigned sort orders
// the four axes by decreasing LSB (ties: zero weight, any order).  STRIDE_SCALE = bytes per LUT entry, folded into
// the strides so the indices are byte offsets.
//
// Instruction budget (gfx950 issues v_and/or/add/sub/lshr in ~2.4 cycles per wave64, every 3-operand or min/max/
// shift-left/24-bit-multiply op in ~4.3: profiles/r01_valu_instruction_rates.txt), per lookup:
//   keys      3 x (v_and + v_lshl_or)                       the centre key is shared by the rotations of a position
//   sort      7 three-input ops  m = max3(a,b,c)  e = med3(a,b,c)  n = min3(a,b,c)
//                                s0 = max(m,d)  s1 = med3(m,e,d)  s2 = med3(e,n,d)  s3 = min(n,d)
//             instead of the 10 of a 5-comparator network
//   indices   the walk ends at base + (all four strides), a constant: vertex 4 is an immediate offset on vertex 0 and
//             vertex 3 = vertex 4 - stride(s3), so only s0, s1, s3 contribute an AND + ADD/SUB (s2 is needed for its
//             LSB alone)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (f << 16) | stride in one instruction (the compiler prefers shift-left + and + or)
__device__ __forceinline__ unsigned make_key(unsigned f, unsigned stride) {
    unsigned r;
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(f), "s"(stride));
    return r;
}

// LDS accesses by 32-bit LDS address.  `smem` is a link-time symbol: pointer arithmetic on it leaves one "+ smem" per
// distinct address chain in the instruction stream (v_add 0); folding the table's LDS address into the walk's base
// index once per position removes them, and constant parts still fold into the DS immediate offset.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;
typedef __attribute__((address_space(3))) const int8_t lds_ci8_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// OFF is applied as a constant element index so that it lands in the DS immediate offset
template <int OFF = 0>
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return ((lds_cu32_t*)a)[OFF / 4]; }
template <int OFF = 0>
__device__ __forceinline__ int lds_ldi8(uint32_t a) { return (int)((lds_ci8_t*)a)[OFF]; }

// A tile pixel is needed twice: its LSB in bits 16..19 of the sort key and its MSB as an index digit.  ds_read_u8_d16_hi
// returns the byte in bits 16..23 (v << 16 for free; what it leaves in the low half is not relied upon), so that
//   key = (r & 0x000F0000) | stride   one v_and_or_b32          msb = r >> 20   one v_lshrrev_b32
// instead of v_and + v_lshl_or + v_lshrrev on a zero-extended byte.  The compiler does not track inline-asm LDS
// loads: pixels_ready() waits for them (it names the registers so that every use is ordered behind it).
__device__ __forceinline__ uint32_t lds_pixel_hi(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
template <int OFF>
__device__ __forceinline__ uint32_t lds_pixel_hi_off(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_u8_d16_hi %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ unsigned key_of(uint32_t r, unsigned stride) {
    unsigned k;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(k) : "v"(r), "s"(0x000F0000u), "v"(stride));
    return k;
}
__device__ __forceinline__ unsigned msb_of(uint32_t r) { return r >> 20; }

// acc + d.lo16 * key.hi16 (signed): one Abel term of a byte-LUT walk, the LSB taken from the sorted key's high half in place --
// sub + mad per term instead of sub + multiply (SDWA) + half a three-input add (round 5: -2 of ~30 instructions per walk)
__device__ __forceinline__ int mad_i16_keyhi(int d, unsigned key, int acc) {
    asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc) : "v"(d), "v"(key));
    return acc;
}

// acc + w.lo16 * d.hi16: the upper half of a packed stage-2 entry multiplied in place (op_sel picks the high half of src1)
__device__ __forceinline__ uint32_t mad_hi16(uint32_t w, uint32_t d, uint32_t acc) {
    asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc) : "v"(w), "v"(d));
    return acc;
}

template <int STRIDE_SCALE>
struct Walk {
    static constexpr int ALL = (kStrideA + kStrideB + kStrideC + kStrideD) * STRIDE_SCALE;   // vertex 4 - vertex 0
    int i0, i1, i2, i3m;        // byte offsets of vertices 0, 1, 2 and (vertex 3 - ALL); vertex 4 = i0 + ALL
    unsigned f0, f1, f2, f3;    // sorted LSBs
    unsigned k0, k1, k2, k3;    // the sorted keys themselves (LSB in the high half: a v_mad_*_i16 reads it there with op_sel)
    // vertex n = address a(n) + constant c(n): the constant goes into the DS immediate offset
    __device__ __forceinline__ uint32_t a(int n) const { return (uint32_t)(n == 0 ? i0 : n == 1 ? i1 : n == 2 ? i2 : n == 3 ? i3m : i0); }
    static constexpr int c(int n) { return n >= 3 ? ALL : 0; }
    __device__ __forceinline__ uint32_t ld32(int n) const { return n >= 3 ? lds_ld32<ALL>(a(n)) : lds_ld32<0>(a(n)); }
    __device__ __forceinline__ int ldi8(int n) const { return n >= 3 ? lds_ldi8<ALL>(a(n)) : lds_ldi8<0>(a(n)); }
};

// ka = key of the centre pixel, basea = its MSB contribution to the index (both shared by the rotations of a position);
// rb, rc, rd = the other three pixels as lds_pixel_hi() returned them; sb, sc, sd = the axis strides (bytes) in VGPRs
template <int STRIDE_SCALE>
__device__ __forceinline__ Walk<STRIDE_SCALE> simplex_walk(unsigned ka, int basea, uint32_t rb, uint32_t rc, uint32_t rd,
                                                           unsigned sb, unsigned sc, unsigned sd) {
    Walk<STRIDE_SCALE> W;
    const unsigned kb = key_of(rb, sb), kc = key_of(rc, sc), kd = key_of(rd, sd);
    // base index, Horner over the MSBs: ((b * 17 + c) * 17 + d) * scale + a-part
    const unsigned t = __umul24(__umul24(msb_of(rb), (unsigned)kL) + msb_of(rc), (unsigned)kL) + msb_of(rd);
    W.i0 = basea + (int)(t * STRIDE_SCALE);
    const unsigned m = umax3(ka, kb, kc), e = umed3(ka, kb, kc), n = umin3(ka, kb, kc);
    const unsigned s0 = m > kd ? m : kd;
    const unsigned s1 = umed3(m, e, kd);
    const unsigned s2 = umed3(e, n, kd);
    const unsigned s3 = n < kd ? n : kd;
    W.i1 = W.i0 + (int)(s0 & 0xFFFFu);
    W.i2 = W.i1 + (int)(s1 & 0xFFFFu);
    W.i3m = W.i0 - (int)(s3 & 0xFFFFu);
    W.f0 = s0 >> 16; W.f1 = s1 >> 16; W.f2 = s2 >> 16; W.f3 = s3 >> 16;
    W.k0 = s0; W.k1 = s1; W.k2 = s2; W.k3 = s3;
    return W;
}

// global -> LDS copy of `bytes` (rounded up to 16) by the whole workgroup; every load of a
// thread is issued before its first LDS write so the L2 latency is paid once per copy
template <int BYTES>
struct Stage16 {
    static constexpr int n = (BYTES + 15) >> 4;
    static constexpr int FULL = n / NT;       // iterations every thread takes part in
    static constexpr int TAIL = n - FULL * NT;
    uint4 r[FULL];
    uint4 rt;
    __device__ __forceinline__ void load(const uint8_t* __restrict__ src, int tid) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
        for (int i = 0; i < FULL; ++i) r[i] = s[tid + i * NT];
        rt = make_uint4(0, 0, 0, 0);
        if (TAIL > 0 && tid < TAIL) rt = s[tid + FULL * NT];
    }
    __device__ __forceinline__ void store(uint8_t* dst, int tid) const {
        uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int i = 0; i < FULL; ++i) d[tid + i * NT] = r[i];
        if (TAIL > 0 && tid < TAIL) d[tid + FULL * NT] = rt;
    }
};

template <int BYTES>
__device__ __forceinline__ void copy16(uint8_t* dst, const uint8_t* __restrict__ src, int tid) {
    Stage16<BYTES> st;
    st.load(src, tid);
    st.store(dst, tid);
}

// One byte-LUT phase over a destination region of NDST px-ch positions (row pitch DP)
// whose centres live in a source tile (pitch SP).  MODE/ROT0/NROT/RSTEP are static so
// the neighbour offsets fold into DS immediates.  PHASE: 0 = first (store), 1 = add,
// 2 = add and finalise with (div, bias) into `dst8`.
// smallest (most negative) neighbour byte offset over the rotations of a phase: DS immediates are unsigned
template <int SP>
__host__ __device__ constexpr int min_tile_offset(char mode, int rot0, int nrot, int rstep) {
    int m = 0;
    for (int i = 0; i < nrot; ++i) {
        const Off3 o = tile_offsets<SP>(mode, rot0 + i * rstep);
        for (int k = 0; k < 3; ++k) m = o.o[k] < m ? o.o[k] : m;
    }
    return m;
}
// the eight neighbours of the 3x3 block around a pixel (byte offsets in a tile of pitch SP) and the index of an offset among them
template <int SP>
__host__ __device__ constexpr int block3_offset(int i) {
    const int dy = i < 3 ? -1 : (i < 5 ? 0 : 1);
    const int dx = i < 3 ? i - 1 : (i < 5 ? (i == 3 ? -1 : 1) : i - 6);
    return dy * SP + dx * CH;
}
template <int SP>
__host__ __device__ constexpr int block3_index(int off) {
    for (int i = 0; i < 8; ++i)
        if (block3_offset<SP>(i) == off) return i;
    return 0;
}
template <int SP, char MODE, int ROT, int MINO>
__device__ __forceinline__ void load_rotation(uint32_t base, uint32_t& rb, uint32_t& rc, uint32_t& rd) {
    constexpr Off3 o = tile_offsets<SP>(MODE, ROT);
    rb = lds_pixel_hi_off<o.o[0] - MINO>(base);
    rc = lds_pixel_hi_off<o.o[1] - MINO>(base);
    rd = lds_pixel_hi_off<o.o[2] - MINO>(base);
}

// stages B-D of a byte-LUT position: NROT simplex walks around the centre pixel ra, all their gathers in flight together,
// and the numerator sum_rot sum_n w_n P_n (weights sum to 16 per lookup)
template <int NROT>
__device__ __forceinline__ int byte_walks(uint32_t lut_a, uint32_t ra, const uint32_t (&rb)[4], const uint32_t (&rc)[4],
                                          const uint32_t (&rd)[4]) {
    const unsigned sa = kStrideA, sb = kStrideB, sc = kStrideC, sd = kStrideD;
    const unsigned ka = key_of(ra, sa);
    const int basea = (int)(__umul24(msb_of(ra), kStrideA) + lut_a);      // LDS address of the base corner
    Walk<1> W[NROT];
    int e[NROT][5];
#ifndef LERF_S1_SPLIT
#define LERF_S1_SPLIT 1
#endif
#if LERF_S1_SPLIT > 0
    // the gathers of a walk (LERF_S1_SPLIT walks) are issued as soon as its indices exist and fly under the next walk (all
    // 4 walks first, then all 20 gathers = LERF_S1_SPLIT 0: 0.8 % slower; groups of two: 0.2 % slower; A/B on one box): the
    // LDS starts on a position's gathers ~100 instructions earlier.  The same split in the stage-2 slot LOST 0.25 %.
#pragma unroll
    for (int h = 0; h < NROT; h += LERF_S1_SPLIT) {
#pragma unroll
        for (int i = h; i < h + LERF_S1_SPLIT; ++i) W[i] = simplex_walk<1>(ka, basea, rb[i], rc[i], rd[i], sb, sc, sd);
#pragma unroll
        for (int i = h; i < h + LERF_S1_SPLIT; ++i)
#pragma unroll
            for (int n = 0; n < 5; ++n) e[i][n] = W[i].ldi8(n);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int i = 0; i < NROT; ++i) W[i] = simplex_walk<1>(ka, basea, rb[i], rc[i], rd[i], sb, sc, sd);
    // stage C: all LUT gathers in flight together
#pragma unroll
    for (int i = 0; i < NROT; ++i)
#pragma unroll
        for (int n = 0; n < 5; ++n) e[i][n] = W[i].ldi8(n);
    __builtin_amdgcn_sched_barrier(0);
#endif
    // stage D: sum_n w_n P_n = 16 P_0 + sum_n f_n (P_{n+1} - P_n)   (w_0 = 16 - f_0, w_n = f_{n-1} - f_n, w_4 = f_3):
    //          four multiply-adds and four subtractions per lookup instead of five weights + five multiply-adds
    int acc = 0, sum0 = 0;
#pragma unroll
    for (int i = 0; i < NROT; ++i) {
        sum0 += e[i][0];
#ifndef LERF_S1_MUL24
        acc = mad_i16_keyhi(e[i][1] - e[i][0], W[i].k0, acc);
        acc = mad_i16_keyhi(e[i][2] - e[i][1], W[i].k1, acc);
        acc = mad_i16_keyhi(e[i][3] - e[i][2], W[i].k2, acc);
        acc = mad_i16_keyhi(e[i][4] - e[i][3], W[i].k3, acc);
#else
        acc += __mul24((int)W[i].f0, e[i][1] - e[i][0]);
        acc += __mul24((int)W[i].f1, e[i][2] - e[i][1]);
        acc += __mul24((int)W[i].f2, e[i][3] - e[i][2]);
        acc += __mul24((int)W[i].f3, e[i][4] - e[i][3]);
#endif
    }
    return acc + kQ * sum0;
}


// NROT (2 or 4) lookups of one byte LUT around the pixel at LDS address `center`; lut_a = LDS address of the LUT.
// Returns the numerator sum_rot sum_n w_n P_n (weights sum to 16 per lookup).
template <int SP, char MODE, int ROT0, int NROT, int RSTEP>
__device__ __forceinline__ int byte_lookups(uint32_t lut_a, uint32_t center) {
    static_assert(NROT == 2 || NROT == 4, "rotation pairs or all four");
    constexpr int MINO = min_tile_offset<SP>(MODE, ROT0, NROT, RSTEP);
    // stage A: every pixel read of the NROT rotations (high-half loads, see lds_pixel_hi)
    const uint32_t base = center + (uint32_t)MINO;
    uint32_t ra = lds_pixel_hi_off<-MINO>(base);
    uint32_t rb[4], rc[4], rd[4];
    if constexpr (MODE == 's' && NROT == 4 && ROT0 == 0 && RSTEP == 1) {
        // the four rotations of the 2x2 pattern cover the 3x3 block around the centre: its four edge neighbours are used
        // by two rotations each, so 8 reads serve the 12 operands
        uint32_t nb[8];
        nb[0] = lds_pixel_hi_off<block3_offset<SP>(0) - MINO>(base); nb[1] = lds_pixel_hi_off<block3_offset<SP>(1) - MINO>(base);
        nb[2] = lds_pixel_hi_off<block3_offset<SP>(2) - MINO>(base); nb[3] = lds_pixel_hi_off<block3_offset<SP>(3) - MINO>(base);
        nb[4] = lds_pixel_hi_off<block3_offset<SP>(4) - MINO>(base); nb[5] = lds_pixel_hi_off<block3_offset<SP>(5) - MINO>(base);
        nb[6] = lds_pixel_hi_off<block3_offset<SP>(6) - MINO>(base); nb[7] = lds_pixel_hi_off<block3_offset<SP>(7) - MINO>(base);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra), "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]), "+v"(nb[4]), "+v"(nb[5]), "+v"(nb[6]), "+v"(nb[7]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const Off3 o = tile_offsets<SP>('s', r);
            rb[r] = nb[block3_index<SP>(o.o[0])];
            rc[r] = nb[block3_index<SP>(o.o[1])];
            rd[r] = nb[block3_index<SP>(o.o[2])];
        }
    } else {
    load_rotation<SP, MODE, ROT0, MINO>(base, rb[0], rc[0], rd[0]);
    load_rotation<SP, MODE, ROT0 + RSTEP, MINO>(base, rb[1], rc[1], rd[1]);
    if constexpr (NROT == 4) {
        load_rotation<SP, MODE, ROT0 + 2 * RSTEP, MINO>(base, rb[2], rc[2], rd[2]);
        load_rotation<SP, MODE, ROT0 + 3 * RSTEP, MINO>(base, rb[3], rc[3], rd[3]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra), "+v"(rb[0]), "+v"(rc[0]), "+v"(rd[0]), "+v"(rb[1]), "+v"(rc[1]), "+v"(rd[1]), "+v"(rb[2]),
                       "+v"(rc[2]), "+v"(rd[2]), "+v"(rb[3]), "+v"(rc[3]), "+v"(rd[3]));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb[0]), "+v"(rc[0]), "+v"(rd[0]), "+v"(rb[1]), "+v"(rc[1]), "+v"(rd[1]));
    }
    }
    return byte_walks<NROT>(lut_a, ra, rb, rc, rd);
}

// The same with the neighbour offsets at run time (general kernels: any of the five sampling patterns).  o[3 i .. 3 i + 2] =
// byte offsets of pixels b, c, d of rotation i in the source tile, wave-uniform (kernel arguments): one address add per
// pixel instead of a DS immediate.
template <int NROT>
__device__ __forceinline__ int byte_lookups_rt(uint32_t lut_a, uint32_t center, const int* __restrict__ o) {
    static_assert(NROT == 2 || NROT == 4, "rotation pairs or all four");
    uint32_t ra = lds_pixel_hi(center);
    uint32_t rb[4], rc[4], rd[4];
#pragma unroll
    for (int i = 0; i < NROT; ++i) {
        rb[i] = lds_pixel_hi(center + (uint32_t)o[3 * i]);
        rc[i] = lds_pixel_hi(center + (uint32_t)o[3 * i + 1]);
        rd[i] = lds_pixel_hi(center + (uint32_t)o[3 * i + 2]);
    }
    if constexpr (NROT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra), "+v"(rb[0]), "+v"(rc[0]), "+v"(rd[0]), "+v"(rb[1]), "+v"(rc[1]), "+v"(rd[1]), "+v"(rb[2]),
                       "+v"(rc[2]), "+v"(rd[2]), "+v"(rb[3]), "+v"(rc[3]), "+v"(rd[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb[0]), "+v"(rc[0]), "+v"(rd[0]), "+v"(rb[1]), "+v"(rc[1]), "+v"(rd[1]));
    return byte_walks<NROT>(lut_a, ra, rb, rc, rd);
}

// position p of a region (pitch DPc px-ch per row, origin (y0g,x0g) in the frame) ->
// byte address of its clamped centre in the source tile (pitch SP, origin (sy0g,sx0g))
template <int DPc, int SP>
__device__ __forceinline__ int center_addr(int p, int y0g, int x0g, int sy0g, int sx0g, int H, int W, bool* inside) {
    int ry = p / DPc;
    int r3 = p - ry * DPc;
    if (H < 0) {                      // interior tile (caller passes H = -1): nothing to clamp
        if (inside) *inside = true;
        return (ry + (y0g - sy0g)) * SP + r3 + (x0g - sx0g) * CH;
    }
    int rx = r3 / CH;
    int c = r3 - rx * CH;
    int gy = y0g + ry, gx = x0g + rx;
    int cy = clampi(gy, 0, H - 1), cx = clampi(gx, 0, W - 1);
    if (inside) *inside = (cy == gy) && (cx == gx);
    return (cy - sy0g) * SP + (cx - sx0g) * CH + c;
}

// SKIP_OUTSIDE: positions outside the frame are not evaluated (callers that never read them: s1_kernel).
// Interior tiles (H < 0, a workgroup-uniform fact) take a loop of their own: a scalar trip count and no clamping or
// frame tests, instead of a per-position interior branch, a vector loop condition and an inside-the-frame mask.
template <int NDST, int DPc, int SP, char MODE, int ROT0, int NROT, int RSTEP, int PHASE>
__device__ __forceinline__ void byte_position(uint32_t lut_a, uint32_t center, int16_t* acc16, uint8_t* dst8, int p, int div, int bias) {
    int v = byte_lookups<SP, MODE, ROT0, NROT, RSTEP>(lut_a, center);
    if (PHASE != 0) v += (int)acc16[p];
    if (PHASE == 2)
        dst8[p] = (uint8_t)rne_div_clip255_fast(v + bias * div, div);
    else
        acc16[p] = (int16_t)v;
}
template <int NDST, int DPc, int SP, char MODE, int ROT0, int NROT, int RSTEP, int PHASE, bool SKIP_OUTSIDE = false>
__device__ __forceinline__ void byte_phase(const int8_t* lut, const uint8_t* src, int16_t* acc16, uint8_t* dst8,
                                           int y0g, int x0g, int sy0g, int sx0g, int H, int W, int div, int bias,
                                           int tid) {
    const uint32_t lut_a = lds_addr(lut), src_a = lds_addr(src);
    if (H < 0) {
        constexpr int FULL = NDST / NT, TAIL = NDST - FULL * NT;
        const uint32_t org = src_a + (uint32_t)((y0g - sy0g) * SP + (x0g - sx0g) * CH);
#pragma unroll 1
        for (int k = 0; k < FULL; ++k) {
            const int p = tid + k * NT;
            const int ry = p / DPc;
            byte_position<NDST, DPc, SP, MODE, ROT0, NROT, RSTEP, PHASE>(lut_a, org + (uint32_t)(ry * (SP - DPc) + p), acc16, dst8, p, div, bias);
        }
        if (TAIL > 0 && tid < TAIL) {
            const int p = tid + FULL * NT;
            const int ry = p / DPc;
            byte_position<NDST, DPc, SP, MODE, ROT0, NROT, RSTEP, PHASE>(lut_a, org + (uint32_t)(ry * (SP - DPc) + p), acc16, dst8, p, div, bias);
        }
        return;
    }
    for (int p = tid; p < NDST; p += NT) {
        bool in = true;
        int a = center_addr<DPc, SP>(p, y0g, x0g, sy0g, sx0g, H, W, SKIP_OUTSIDE ? &in : nullptr);
        if (SKIP_OUTSIDE && !in) continue;
        byte_position<NDST, DPc, SP, MODE, ROT0, NROT, RSTEP, PHASE>(lut_a, src_a + (uint32_t)a, acc16, dst8, p, div, bias);
    }
}

// ---------------------------------------------------------------------------
// Stage-1 positions of INTERIOR tiles with the pixel reads on the vector-memory path (s1_kernel): stage 1 is bound by the
// LDS (the byte-LUT gathers keep its array 82 % busy, PMC) while the texture path idles -- so the 9 / 13 neighbourhood
// pixels of a position are fetched straight from the frame (global_load_ubyte_d16_hi: the same "byte in bits 16..23" form
// the LDS reads deliver; L1 / L2 hits, the tile's rows are read 13 times over), one position AHEAD of the lookups, and the
// LDS serves the LUT gathers and the partial sums only.  Addresses cost nothing: address = row base (one SGPR pair per
// dy in -3..3) + the position's byte offset (one VGPR) + dx * CH (immediate).
// ---------------------------------------------------------------------------
struct DyDx { int dy[3], dx[3]; };
__host__ __device__ constexpr DyDx pattern_dydx(char mode, int rot) {
    int dy[4] = {0, 0, 0, 0}, dx[4] = {0, 0, 0, 0};
    switch (mode) {
        case 's': dy[1] = 0; dx[1] = 1; dy[2] = 1; dx[2] = 0; dy[3] = 1; dx[3] = 1; break;
        case 'd': dy[1] = 0; dx[1] = 2; dy[2] = 2; dx[2] = 0; dy[3] = 2; dx[3] = 2; break;
        case 'y': dy[1] = 1; dx[1] = 1; dy[2] = 1; dx[2] = 2; dy[3] = 2; dx[3] = 1; break;
        case 'c': dy[1] = 0; dx[1] = 1; dy[2] = 0; dx[2] = 2; dy[3] = 0; dx[3] = 3; break;
        default:  dy[1] = 1; dx[1] = 1; dy[2] = 2; dx[2] = 2; dy[3] = 3; dx[3] = 3; break;   // 't'
    }
    DyDx r{};
    for (int k = 1; k < 4; ++k) {
        int y = dy[k], x = dx[k];
        for (int i = 0; i < (rot & 3); ++i) { int t = y; y = x; x = -t; }
        r.dy[k - 1] = y;
        r.dx[k - 1] = x;
    }
    return r;
}
template <int IMM>
__device__ __forceinline__ uint32_t gl_pixel_hi(uint32_t voff, uint64_t rowbase) {
    uint32_t r;
    asm volatile("global_load_ubyte_d16_hi %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(rowbase), "n"(IMM));
    return r;
}
struct PixSet { uint32_t a, b[4], c[4], d[4]; };
// rows[dy + 3] = address of the region's first centre pixel, dy rows away; rotations ROT0 + i * RSTEP, i < NROT
template <char MODE, int ROT0, int NROT, int RSTEP>
__device__ __forceinline__ void pix_issue(PixSet& P, uint32_t voff, const uint64_t (&rows)[7]) {
    P.a = gl_pixel_hi<0>(voff, rows[3]);
#define LERF_PX(I, K) gl_pixel_hi<pattern_dydx(MODE, ROT0 + (I) * RSTEP).dx[K] * CH>(voff, rows[pattern_dydx(MODE, ROT0 + (I) * RSTEP).dy[K] + 3])
    if constexpr (MODE == 's' && NROT == 4 && ROT0 == 0 && RSTEP == 1) {
        // the four rotations of the 2x2 pattern cover the 3x3 block: its edge neighbours E, S, W, N are pixel b of one
        // rotation and pixel c of the previous one -- 8 loads serve the 12 operands (pix_ready hands them out)
        P.b[0] = LERF_PX(0, 0); P.b[1] = LERF_PX(1, 0); P.b[2] = LERF_PX(2, 0); P.b[3] = LERF_PX(3, 0);
        P.d[0] = LERF_PX(0, 2); P.d[1] = LERF_PX(1, 2); P.d[2] = LERF_PX(2, 2); P.d[3] = LERF_PX(3, 2);
    } else {
        P.b[0] = LERF_PX(0, 0); P.c[0] = LERF_PX(0, 1); P.d[0] = LERF_PX(0, 2);
        P.b[1] = LERF_PX(1, 0); P.c[1] = LERF_PX(1, 1); P.d[1] = LERF_PX(1, 2);
        if constexpr (NROT == 4) {
            P.b[2] = LERF_PX(2, 0); P.c[2] = LERF_PX(2, 1); P.d[2] = LERF_PX(2, 2);
            P.b[3] = LERF_PX(3, 0); P.c[3] = LERF_PX(3, 1); P.d[3] = LERF_PX(3, 2);
        }
    }
#undef LERF_PX
}
template <char MODE, int ROT0, int NROT, int RSTEP>
__device__ __forceinline__ void pix_ready(PixSet& P) {
    if constexpr (MODE == 's' && NROT == 4 && ROT0 == 0 && RSTEP == 1) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(P.a), "+v"(P.b[0]), "+v"(P.b[1]), "+v"(P.b[2]), "+v"(P.b[3]), "+v"(P.d[0]), "+v"(P.d[1]),
                     "+v"(P.d[2]), "+v"(P.d[3]));
        static_assert(pattern_dydx('s', 0).dy[1] == pattern_dydx('s', 1).dy[0] && pattern_dydx('s', 0).dx[1] == pattern_dydx('s', 1).dx[0],
                      "pixel c of rotation r = pixel b of rotation r + 1");
        P.c[0] = P.b[1]; P.c[1] = P.b[2]; P.c[2] = P.b[3]; P.c[3] = P.b[0];
    } else if constexpr (NROT == 4) {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(P.a), "+v"(P.b[0]), "+v"(P.c[0]), "+v"(P.d[0]), "+v"(P.b[1]), "+v"(P.c[1]), "+v"(P.d[1]), "+v"(P.b[2]),
                       "+v"(P.c[2]), "+v"(P.d[2]), "+v"(P.b[3]), "+v"(P.c[3]), "+v"(P.d[3]));
    } else {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(P.a), "+v"(P.b[0]), "+v"(P.c[0]), "+v"(P.d[0]), "+v"(P.b[1]), "+v"(P.c[1]), "+v"(P.d[1]));
    }
}
// one byte-LUT phase over the NDST positions (row pitch DPc) of an INTERIOR region; origin = address of the region's first
// centre pixel in the frame (stage 1: the input, stage 2 of LeRF-L: the stage-1 output), pitch = bytes per frame row.
// PHASE as in byte_position.  Threads past the last position of the final round repeat it (loads only).
template <int NDST, int DPc, char MODE, int ROT0, int NROT, int RSTEP, int PHASE>
__device__ __forceinline__ void byte_phase_vmem(const int8_t* lut, const uint8_t* __restrict__ origin, int pitch, int16_t* acc16,
                                                uint8_t* dst8, int div, int bias, int tid) {
    constexpr int ROUNDS = (NDST + NT - 1) / NT, PAIRS = (ROUNDS + 1) / 2;
    const uint32_t lut_a = lds_addr(lut);
    uint64_t rows[7];
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy) rows[dy + 3] = (uint64_t)reinterpret_cast<uintptr_t>(origin + (int64_t)dy * pitch);
    auto voff_of = [&](int p) {
        p = p < NDST ? p : NDST - 1;
        const int ry = p / DPc;
        return (uint32_t)(ry * pitch + (p - ry * DPc));
    };
    auto finish = [&](const PixSet& X, int p) {
        if (NDST % NT != 0 && p >= NDST) return;
        int v = byte_walks<NROT>(lut_a, X.a, X.b, X.c, X.d);
        if (PHASE != 0) v += (int)acc16[p];
        if (PHASE == 2)
            dst8[p] = (uint8_t)rne_div_clip255_fast(v + bias * div, div);
        else
            acc16[p] = (int16_t)v;
    };
    PixSet A, B;
    pix_issue<MODE, ROT0, NROT, RSTEP>(A, voff_of(tid), rows);
#pragma unroll 1
    for (int k = 0; k < PAIRS; ++k) {
        const int p = tid + 2 * k * NT;
        pix_ready<MODE, ROT0, NROT, RSTEP>(A);
        pix_issue<MODE, ROT0, NROT, RSTEP>(B, voff_of(p + NT), rows);
        finish(A, p);
        pix_ready<MODE, ROT0, NROT, RSTEP>(B);
        if (k + 1 < PAIRS) pix_issue<MODE, ROT0, NROT, RSTEP>(A, voff_of(p + 2 * NT), rows);
        if (2 * k + 1 < ROUNDS) finish(B, p + NT);
    }
}

// The same for the general kernels: pattern offsets at run time.  dyx[3 i + k] = (dy << 16) | (dx & 0xFFFF) of pixel k + 1 of
// rotation i (frame coordinates, wave-uniform kernel arguments); a pixel's address = origin + (position offset + dy * pitch
// + dx * CH): one VALU add per pixel, like the LDS path of byte_lookups_rt.
__device__ __forceinline__ uint32_t gl_pixel_hi_rt(uint32_t voff, uint64_t origin) {
    uint32_t r;
    asm volatile("global_load_ubyte_d16_hi %0, %1, %2" : "=v"(r) : "v"(voff), "s"(origin));
    return r;
}
template <int NDST, int DPc, int NROT>
__device__ __forceinline__ void byte_phase_vmem_rt(const int8_t* lut, const uint8_t* __restrict__ origin, int pitch, int16_t* acc16,
                                                   uint8_t* dst8, int div, int bias, int tid, const int* __restrict__ dyx, bool first,
                                                   bool last) {
    constexpr int ROUNDS = (NDST + NT - 1) / NT, PAIRS = (ROUNDS + 1) / 2;
    const uint32_t lut_a = lds_addr(lut);
    // the VGPR offset of a global load is UNSIGNED: the base is moved up-left by the largest pattern reach (3 rows, 3 pixels;
    // inside the frame for an interior region) so that every pixel offset is non-negative
    const int reach = 3 * pitch + 3 * CH;
    const uint64_t org = (uint64_t)reinterpret_cast<uintptr_t>(origin - reach);
    int off[NROT * 3];
#pragma unroll
    for (int i = 0; i < NROT * 3; ++i) off[i] = (dyx[i] >> 16) * pitch + (int)(int16_t)(dyx[i] & 0xFFFF) * CH;
    auto voff_of = [&](int p) {
        p = p < NDST ? p : NDST - 1;
        const int ry = p / DPc;
        return (uint32_t)(ry * pitch + (p - ry * DPc) + reach);
    };
    auto issue = [&](PixSet& X, uint32_t voff) {
        X.a = gl_pixel_hi_rt(voff, org);
#pragma unroll
        for (int i = 0; i < NROT; ++i) {
            X.b[i] = gl_pixel_hi_rt(voff + (uint32_t)off[3 * i], org);
            X.c[i] = gl_pixel_hi_rt(voff + (uint32_t)off[3 * i + 1], org);
            X.d[i] = gl_pixel_hi_rt(voff + (uint32_t)off[3 * i + 2], org);
        }
    };
    auto ready = [&](PixSet& X) {
        if constexpr (NROT == 4)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(X.a), "+v"(X.b[0]), "+v"(X.c[0]), "+v"(X.d[0]), "+v"(X.b[1]), "+v"(X.c[1]), "+v"(X.d[1]), "+v"(X.b[2]),
                           "+v"(X.c[2]), "+v"(X.d[2]), "+v"(X.b[3]), "+v"(X.c[3]), "+v"(X.d[3]));
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(X.a), "+v"(X.b[0]), "+v"(X.c[0]), "+v"(X.d[0]), "+v"(X.b[1]), "+v"(X.c[1]), "+v"(X.d[1]));
    };
    auto finish = [&](const PixSet& X, int p) {
        if (NDST % NT != 0 && p >= NDST) return;
        int v = byte_walks<NROT>(lut_a, X.a, X.b, X.c, X.d);
        if (!first) v += (int)acc16[p];
        if (last)
            dst8[p] = (uint8_t)rne_div_clip255_fast(v + bias * div, div);
        else
            acc16[p] = (int16_t)v;
    };
    PixSet A, B;
    issue(A, voff_of(tid));
#pragma unroll 1
    for (int k = 0; k < PAIRS; ++k) {
        const int p = tid + 2 * k * NT;
        ready(A);
        issue(B, voff_of(p + NT));
        finish(A, p);
        ready(B);
        if (k + 1 < PAIRS) issue(A, voff_of(p + 2 * NT));
        if (2 * k + 1 < ROUNDS) finish(B, p + NT);
    }
}

// General kernels: the same phase with the pattern offsets `o` (see byte_lookups_rt) and the phase kind at run time --
// `first`: store the partial sums, otherwise add; `last`: finalise with (div, bias) into dst8.  All wave-uniform.
template <int NROT>
__device__ __forceinline__ void byte_position_rt(uint32_t lut_a, uint32_t center, int16_t* acc16, uint8_t* dst8, int p, int div,
                                                 int bias, const int* __restrict__ o, bool first, bool last) {
    int v = byte_lookups_rt<NROT>(lut_a, center, o);
    if (!first) v += (int)acc16[p];
    if (last)
        dst8[p] = (uint8_t)rne_div_clip255_fast(v + bias * div, div);
    else
        acc16[p] = (int16_t)v;
}
template <int NDST, int DPc, int SP, int NROT, bool SKIP_OUTSIDE = false>
__device__ __forceinline__ void byte_phase_rt(const int8_t* lut, const uint8_t* src, int16_t* acc16, uint8_t* dst8, int y0g, int x0g,
                                              int sy0g, int sx0g, int H, int W, int div, int bias, int tid,
                                              const int* __restrict__ o, bool first, bool last) {
    const uint32_t lut_a = lds_addr(lut), src_a = lds_addr(src);
    if (H < 0) {
        const uint32_t org = src_a + (uint32_t)((y0g - sy0g) * SP + (x0g - sx0g) * CH);
#pragma unroll 1
        for (int p = tid; p < NDST; p += NT) {
            const int ry = p / DPc;
            byte_position_rt<NROT>(lut_a, org + (uint32_t)(ry * (SP - DPc) + p), acc16, dst8, p, div, bias, o, first, last);
        }
        return;
    }
#pragma unroll 1
    for (int p = tid; p < NDST; p += NT) {
        bool in = true;
        int a = center_addr<DPc, SP>(p, y0g, x0g, sy0g, sx0g, H, W, SKIP_OUTSIDE ? &in : nullptr);
        if (SKIP_OUTSIDE && !in) continue;
        byte_position_rt<NROT>(lut_a, src_a + (uint32_t)a, acc16, dst8, p, div, bias, o, first, last);
    }
}

// ---------------------------------------------------------------------------
// stage 3 helpers
// ---------------------------------------------------------------------------
// first i in [0,n) with a[i] >= key (a is non-decreasing), all 64 lanes of a wave cooperating:
// 64-way splits, so a table of 8k entries needs three dependent loads instead of thirteen
__device__ __forceinline__ int wave_lower_bound(const int* __restrict__ a, int n, int key, int lane) {
    int lo = 0, hi = n;
    while (hi - lo > 64) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + lane * step;
        const bool less = idx < hi && a[idx] < key;
        const int cnt = __popcll(__ballot(less));          // monotone: the first cnt probes are < key
        const int nlo = cnt > 0 ? lo + (cnt - 1) * step + 1 : lo;
        const int nhi = cnt < 64 ? min(hi, lo + cnt * step) : hi;
        lo = nlo;
        hi = nhi;
    }
    const int idx = lo + lane;
    const bool less = idx < hi && a[idx] < key;
    return lo + __popcll(__ballot(less));
}


// ---- input tile -> LDS.  Interior tiles whose rows all start at the same offset from a 4-byte boundary (frame pitch
//      and frame stride multiples of 4: every RGB frame whose width is a multiple of 4) are fetched as aligned dwords
//      and land in LDS with that offset as a phase (returned); the other tiles go byte by byte with clamped
//      coordinates (np.pad(..., 'edge') in every rotated frame).  Every load of a thread is issued before its first
//      LDS store (one L2/HBM latency); `between()` runs while the loads are in flight.
template <int IY, int IPB, int IP, typename F>
__device__ __forceinline__ int load_input_tile(uint8_t* Ct, const uint8_t* __restrict__ img, int H, int W, int iy0,
                                               int ix0, bool interior, int tid, F between) {
    int cphase = 0;
    const int64_t row0 = ((int64_t)iy0 * W + ix0) * CH;                     // first byte of the region in the frame
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(img) + (uintptr_t)row0;
    const bool dwords = interior && ((W * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0 &&
                        (iy0 + IY < H || (ix0 * CH - (int)(a0 & 3)) + IP <= W * CH);   // never read past the frame
    if (dwords) {
        cphase = (int)(a0 & 3);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a0 - (uintptr_t)cphase);
        constexpr int RD = IP / 4, ND = IY * RD, KD = (ND + NT - 1) / NT;
        const int rowdw = (W * CH) >> 2;
        uint32_t v[KD];
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int p = min(tid + k * NT, ND - 1);
            const int ry = p / RD;
            v[k] = __builtin_nontemporal_load(&src[(int64_t)ry * rowdw + (p - ry * RD)]);   // read once: leave the L2 to the LUT pack
        }
        between();
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int p = tid + k * NT;
            if (p < ND) reinterpret_cast<uint32_t*>(Ct)[p] = v[k];
        }
    } else {
        constexpr int NB = IY * IPB, KI = (NB + NT - 1) / NT;
        uint8_t v[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int p = min(tid + k * NT, NB - 1);
            int ry = p / IPB;
            int r3 = p - ry * IPB;
            int rx = r3 / CH;
            int c = r3 - rx * CH;
            int gy = clampi(iy0 + ry, 0, H - 1), gx = clampi(ix0 + rx, 0, W - 1);
            v[k] = img[((int64_t)gy * W + gx) * CH + c];
        }
        between();
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int p = tid + k * NT;
            if (p < NB) {
                const int ry = p / IPB;
                Ct[ry * IP + (p - ry * IPB)] = v[k];
            }
        }
    }
    return cphase;
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8; an affinity, not a guarantee -- nothing
// depends on it but speed).  Handing every XCD a CONTIGUOUS eighth of the (frame, tile) sequence instead of every eighth tile
// keeps the tiles that run side by side on one XCD neighbours in the frame: their input / feat halos and the LUT pieces
// they ask for at the same time meet in that XCD's own L2.  Measured: L2 hit rate 94.8 -> 97.5 %, HBM fetch of the two launches
// 169 -> 73 MB per step, +1.3 % throughput.  A bijection of [0, total) for any total.
// Small launches (under four rounds of workgroups) keep the linear order: their cheap partial tiles at the end of the
// sequence would all land on the last XCD (a 300-tile strip launch took 0.35 instead of 0.27 ms).
__device__ __forceinline__ int xcd_order(int b, int total) {
    if (total < 1024) return b;
    const int x = b & 7, j = b >> 3, q = total >> 3, r = total & 7;
    return x * q + (x < r ? x : r) + j;
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
// EMIT = false: the whole SR path.  EMIT = true: stages 1+2 only; the tile's own 64x64 block of
// (hq0,hq1,hq2,feat) dwords goes to P.emit ([H][W][3] uint32) for the warp kernels / the stage API.
// FROM_FEAT = true: stage 1 has been run by s1_kernel over the whole batch; the feat tile (with its halo) is read
// from P.feat instead of being recomputed from the input tile -- stage 1 then costs 64x64 instead of 72x72 positions
// per tile (-21 % of its work) for 6 bytes of extra HBM traffic per LR pixel.
// GEN = true: the general kernels -- any 1..4 sampling patterns per stage (run-time neighbour offsets), scale factors up to
// x8 (large geometry tables, staged late), frames of different sizes in one launch (RaggedTable).  GEN = false: the
// specialised kernels of the published configuration (modes "sct" / "sct", scale < 4.9, one frame size).
// What a workgroup needs to know about ITS frame, as plain values (never a modified copy of the kernel arguments: a
// 600-byte struct that is written through a reference ends up in scratch memory and every later P.x becomes a scratch load --
// the first version of the general kernels ran at half speed for that reason).
struct FrameView {
    const uint8_t* img; uint8_t* out; uint8_t* feat; uint32_t* emit;
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;
    int H, W, oH, oW, tiles_x, tiles_y;
};
template <bool GEN>
__device__ __forceinline__ FrameView frame_view(const Params& P, const std::conditional_t<GEN, RaggedTable, NoTable>& T, int& bid) {
    FrameView F;
    if constexpr (GEN) {
        if (T.n > 0) {
            int f = 0;
            for (int k = 1; k < T.n; ++k)
                if (bid >= T.d[k].first_block) f = k;
            bid -= T.d[f].first_block;
            F.img = T.d[f].img; F.out = T.d[f].out; F.feat = T.d[f].feat; F.emit = T.d[f].emit;
            F.left_r = T.d[f].left_r; F.dis_r = T.d[f].dis_r; F.left_c = T.d[f].left_c; F.dis_c = T.d[f].dis_c;
            F.dis_r64 = T.d[f].dis_r64; F.dis_c64 = T.d[f].dis_c64;
            F.H = T.d[f].H; F.W = T.d[f].W; F.oH = T.d[f].oH; F.oW = T.d[f].oW;
            F.tiles_x = T.d[f].tiles_x; F.tiles_y = T.d[f].tiles_y;
            return F;
        }
    }
    const int tiles = P.tiles_y * P.tiles_x;
    const int frame = bid / tiles;
    bid -= frame * tiles;
    F.img = P.img + frame * P.in_sn; F.out = P.out + frame * P.out_sn;
    F.feat = P.feat + frame * P.feat_sn; F.emit = P.emit + frame * P.emit_sn;
    F.left_r = P.left_r; F.dis_r = P.dis_r; F.left_c = P.left_c; F.dis_c = P.dis_c;
    F.dis_r64 = P.dis_r64; F.dis_c64 = P.dis_c64;
    F.H = P.H; F.W = P.W; F.oH = P.oH; F.oW = P.oW; F.tiles_x = P.tiles_x; F.tiles_y = P.tiles_y;
    return F;
}

// The warp of one source tile (tile-fused warp, lerf_warp_fused_u8): every output pixel of the tile's box is projected (float64,
// Warp2dNumpy's operation order) and kept when this tile owns it (host::warp_owner_key: last tap row / column inside the tile);
// its 2 x 2 taps are packed dwords (hq0 | hq1 << 8 | hq2 << 16 | feat << 24) of the tile's stage-2 region Dt (origin
// (hy0, hx0), pitch HP dwords) -- the per-pixel arithmetic is warp_px_value_u8, the packed warp kernel's own.
template <int KIND, int HP>
__device__ __forceinline__ void warp_tile(const Params& P, const uint32_t* Dt, int hy0, int hx0, int ty0, int tx0, int H, int W, int tile,
                                          uint8_t* __restrict__ outp, int tid) {
    const WarpGeo& g = P.wgeo;
    const int i0 = P.wboxes[4 * tile], i1 = P.wboxes[4 * tile + 1], j0 = P.wboxes[4 * tile + 2], j1 = P.wboxes[4 * tile + 3];
    const int bw = j1 - j0, n = (i1 - i0) * bw;
    if (n <= 0) return;
    const float gsc = KIND == LERF_KIND_GAUSS ? s3::gauss_scale(P.max_sigma) : 1.0f;
    const unsigned inv_bw = (unsigned)((0xFFFFFFFFull / (unsigned)bw) + 1ull);          // q / bw for q < 2^26 (checked by the launcher)
    for (int q = tid; q < n; q += NT) {
        int di = (int)__umulhi((unsigned)q, inv_bw);
        int dj = q - di * bw;
        if (dj < 0) { --di; dj += bw; } else if (dj >= bw) { ++di; dj -= bw; }
        const int i = i0 + di, j = j0 + dj;
        const WarpPx2 G = warp_px_geometry(g, i, j, H, W);
        const int kr = min(G.rrow[0] + 1, H - 1), kc = min(G.rcol[0] + 1, W - 1);
        if (kr < ty0 || kr >= ty0 + TH || kc < tx0 || kc >= tx0 + TW) continue;
        const float dxs[2] = {G.dx[0] * gsc, G.dx[1] * gsc}, dys[2] = {G.dy[0] * gsc, G.dy[1] * gsc};
        uint8_t* dst = outp + ((int64_t)i * g.oW + j) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            auto tap = [&](int r, int cc) -> uint32_t { return Dt[(r - hy0) * HP + (cc - hx0) * CH + c]; };
            float res;
            if (warp_px_value_u8<KIND>(G, g, H, W, P.max_sigma, dxs, dys, tap, dst + c, &res)) continue;
            dst[c] = s3::to_u8(res);
        }
    }
}

// KINDW = LERF_KIND_* (| LERF_FUSED_WARP: stage 3 is the homographic warp of the tile's own output pixels, warp_tile())
constexpr int LERF_FUSED_WARP = 16;
template <int S, int KINDW, bool EMIT, bool FROM_FEAT = false, bool GEN = false>
__global__ void __launch_bounds__(NT)
sr_fused_kernel(Params P, std::conditional_t<GEN, RaggedTable, NoTable> T) {
    constexpr int KIND = KINDW & 15;
    constexpr bool WARP = (KINDW & LERF_FUSED_WARP) != 0;
    static_assert(!WARP || (!EMIT && S == 2 && CH == 3 && !GEN), "tile-fused warp: RGB, 2 x 2 support, specialised kernel");
    using D = Dims<S, GEN, EMIT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    int bid = xcd_order((int)blockIdx.x, (int)gridDim.x);
    const FrameView F = frame_view<GEN>(P, T, bid);
    const int tyi = bid / F.tiles_x, txi = bid - tyi * F.tiles_x;
    const int ty0 = P.ty_org + tyi * TH, tx0 = P.tx_org + txi * TW;
    const int H = F.H, W = F.W;
    const uint8_t* __restrict__ img = F.img;
    uint8_t* __restrict__ outp = F.out;

    // region origins in frame coordinates
    const int hy0 = ty0 - D::HR, hx0 = tx0 - D::HR;
    const int fy0 = ty0 - D::R3 - R2, fx0 = tx0 - D::R3 - R2;
    const int iy0 = fy0 - R1, ix0 = fx0 - R1;

    uint8_t* Bt = smem + D::OFF_B;
    int* ctl = reinterpret_cast<int*>(smem + D::OFF_CTL);
    // interior tile: the whole input region lies inside the frame, so no coordinate is ever clamped and the
    // centre addresses reduce to a multiply-add (center_addr's H < 0 path); ~80 % of the tiles of a 1080p frame
    const bool interior = FROM_FEAT ? (fy0 >= 0 && fx0 >= 0 && fy0 + D::FY <= F.H && fx0 + D::FX <= F.W)
                                    : (iy0 >= 0 && ix0 >= 0 && iy0 + D::IY <= F.H && ix0 + D::IX <= F.W);
    const int Hc = interior ? -1 : F.H, Wc = F.W;

    int* g_lr = reinterpret_cast<int*>(smem + D::OFF_GEO);
    float* g_dr = reinterpret_cast<float*>(g_lr + D::GEO_ROWS);
    int* g_lc = reinterpret_cast<int*>(g_dr + D::GEO_ROWS * S);
    float* g_dc = reinterpret_cast<float*>(g_lc + D::GEO_COLS);
    // owned output rows / columns: four wave-parallel searches in the global tables
    auto geo_search = [&]() {
        if (wave < 4) {
            const bool rows = wave < 2;
            const int* tab = rows ? F.left_r : F.left_c;
            const int n = rows ? F.oH : F.oW;
            const int ti = rows ? tyi : txi, tn = rows ? F.tiles_y : F.tiles_x, t0 = rows ? ty0 : tx0;
            int r;
            if (!(wave & 1)) r = ti == 0 ? 0 : wave_lower_bound(tab, n, t0 - D::R3, lane);
            else r = ti == tn - 1 ? n : wave_lower_bound(tab, n, t0 + (rows ? TH : TW) - D::R3, lane);
            if (lane == 0) ctl[16 + wave] = r;
        }
    };
    // tables of the owned block into LDS; LeRF-G distances pre-multiplied by (max_sigma/255)*sqrt(0.5 log2 e)
    auto geo_stage = [&]() {
        const int gi0 = ctl[16], gi1 = ctl[17], gj0 = ctl[18], gj1 = ctl[19];
        const float gscale = KIND == LERF_KIND_GAUSS ? s3::gauss_scale(P.max_sigma) : 1.0f;
        for (int e = tid; e < gi1 - gi0; e += NT) {
            g_lr[e] = F.left_r[gi0 + e] - hy0;
#pragma unroll
            for (int b = 0; b < S; ++b) g_dr[e * S + b] = F.dis_r[(gi0 + e) * S + b] * gscale;
        }
        for (int e = tid; e < gj1 - gj0; e += NT) {
            g_lc[e] = F.left_c[gj0 + e] - hx0;
#pragma unroll
            for (int a = 0; a < S; ++a) g_dc[e * S + a] = F.dis_c[(gj0 + e) * S + a] * gscale;
        }
    };

    LERF_STAMP_RT(13);
    LERF_STAMP(0);
#ifdef LERF_STAMPS
    if (tid == 0) { P.stamps[(size_t)blockIdx.x * 16 + 8] = 0; P.stamps[(size_t)blockIdx.x * 16 + 9] = 0; P.stamps[(size_t)blockIdx.x * 16 + 15] = 0; }
#endif
    if (!FROM_FEAT) {
    // ---- input tile (load_input_tile), the geometry search riding behind its loads.  (The vector-memory pixel path of
    //      s1_kernel was tried here too -- 72 x 72 positions per tile -- and did not pay: 0.191 vs 0.184 ms per 1080 x 960
    //      block; this kernel's stage 1 shares the CU with nothing else that uses the LDS less.)
    const int cphase = load_input_tile<D::IY, D::IPB, D::IP>(smem + D::OFF_C, img, H, W, iy0, ix0, interior, tid,
                                                             [&]() { if (D::GEO_EARLY && !EMIT && !WARP) geo_search(); });

    // ---- stage 1: three byte LUTs, 4 rotations each (eval_lut_sr.py:541-577)
    {
        const uint8_t* Ct = smem + D::OFF_C + cphase;
        int8_t* lut = reinterpret_cast<int8_t*>(smem + D::OFF_LUT);
        int16_t* acc = reinterpret_cast<int16_t*>(smem + D::OFF_ACC);
        const int div1 = kQ * 3;
        copy16<LERF_LUT_ENTRIES>(smem + D::OFF_LUT, P.pack + 0 * LUT_PAD, tid);
        __syncthreads();
        if (D::GEO_EARLY && !EMIT && !WARP) geo_stage();            // search results are in ctl; its loads hide behind phase s
        LERF_STAMP(1);
        // the next LUT rides in registers behind the lookups (explicit scalars: an array/struct that lives
        // across the position loop ends up in scratch)
        constexpr int L1N = (LERF_LUT_ENTRIES + 15) / 16, L1TAIL = L1N - 5 * NT;
        static_assert(L1TAIL > 0 && L1TAIL <= NT, "stage-1 LUT = 5 full uint4 rounds + a tail");
        uint4 n0, n1, n2, n3, n4, n5 = make_uint4(0, 0, 0, 0);
#define LERF_S1_LOAD(SRC)                                                                          \
        do {                                                                                       \
            const uint4* s_ = reinterpret_cast<const uint4*>(SRC);                                 \
            n0 = s_[tid]; n1 = s_[tid + NT]; n2 = s_[tid + 2 * NT]; n3 = s_[tid + 3 * NT]; n4 = s_[tid + 4 * NT]; \
            if (tid < L1TAIL) n5 = s_[tid + 5 * NT];                                               \
        } while (0)
#define LERF_S1_STORE()                                                                            \
        do {                                                                                       \
            uint4* d_ = reinterpret_cast<uint4*>(smem + D::OFF_LUT);                               \
            d_[tid] = n0; d_[tid + NT] = n1; d_[tid + 2 * NT] = n2; d_[tid + 3 * NT] = n3; d_[tid + 4 * NT] = n4; \
            if (tid < L1TAIL) d_[tid + 5 * NT] = n5;                                               \
        } while (0)
        if constexpr (GEN) {
            // any 1..4 patterns: the same phases with run-time offsets (lerf_luts_t.modes1 order)
            const int dv = kQ * P.n1;
#pragma unroll 1
            for (int m = 0; m < P.n1; ++m) {
                const bool more = m + 1 < P.n1;
                if (more) LERF_S1_LOAD(P.pack + (size_t)(m + 1) * LUT_PAD);
                byte_phase_rt<D::NF, D::FP, D::IP, 4>(lut, Ct, acc, Bt, fy0, fx0, iy0, ix0, Hc, Wc, dv, 0, tid, P.s1off[m], m == 0, !more);
                __syncthreads();
                if (more) {
                    LERF_S1_STORE();
                    __syncthreads();
                }
            }
        } else {
        LERF_S1_LOAD(P.pack + 1 * LUT_PAD);
        byte_phase<D::NF, D::FP, D::IP, 's', 0, 4, 1, 0>(lut, Ct, acc, Bt, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        LERF_STAMP(2);
        LERF_S1_STORE();
        __syncthreads();
        LERF_STAMP(3);
        LERF_S1_LOAD(P.pack + 2 * LUT_PAD);
        byte_phase<D::NF, D::FP, D::IP, 'c', 0, 4, 1, 1>(lut, Ct, acc, Bt, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        LERF_STAMP(4);
        LERF_S1_STORE();
        __syncthreads();
        LERF_STAMP(5);
        byte_phase<D::NF, D::FP, D::IP, 't', 0, 4, 1, 2>(lut, Ct, acc, Bt, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        LERF_STAMP(6);
        }
    }
    } else {
        // ---- feat tile (with its halo) from the stage-1 launch; out-of-frame positions take the clamped pixel,
        //      which is what stage 1 evaluates for them
        const uint8_t* __restrict__ fsrc = F.feat;
        const bool dwords = interior && (D::FP & 3) == 0 && ((W * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(F.feat) & 3) == 0 && ((fx0 * CH) & 3) == 0;
        if (dwords) {
            constexpr int RD = D::FP / 4, ND = D::FY * RD, KD = (ND + NT - 1) / NT;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(fsrc + ((int64_t)fy0 * W + fx0) * CH);
            const int rowdw = (W * CH) >> 2;
            uint32_t v[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const int p = min(tid + k * NT, ND - 1);
                const int ry = p / RD;
                v[k] = __builtin_nontemporal_load(&src[(int64_t)ry * rowdw + (p - ry * RD)]);
            }
            if (D::GEO_EARLY && !EMIT && !WARP) geo_search();
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const int p = tid + k * NT;
                if (p < ND) reinterpret_cast<uint32_t*>(Bt)[p] = v[k];
            }
        } else {
            constexpr int KI = (D::NF + NT - 1) / NT;
            uint8_t v[KI];
#pragma unroll
            for (int k = 0; k < KI; ++k) {
                const int p = min(tid + k * NT, D::NF - 1);
                const int ry = p / D::FP;
                const int r3 = p - ry * D::FP;
                const int rx = r3 / CH;
                const int gy = clampi(fy0 + ry, 0, H - 1), gx = clampi(fx0 + rx, 0, W - 1);
                v[k] = fsrc[((int64_t)gy * W + gx) * CH + (r3 - rx * CH)];
            }
            if (D::GEO_EARLY && !EMIT && !WARP) geo_search();
#pragma unroll
            for (int k = 0; k < KI; ++k) {
                const int p = tid + k * NT;
                if (p < D::NF) Bt[p] = v[k];
            }
        }
        __syncthreads();
        if (D::GEO_EARLY && !EMIT && !WARP) geo_stage();
        LERF_STAMP(6);
    }

    uint32_t* Dt = reinterpret_cast<uint32_t*>(smem + D::OFF_D);

    // image byte stage 3 sees at a hyper-region position OUTSIDE the frame: np.pad(input, pad_vec, mode=self.pad_mode)
    // (resize_right2d_numpy.py:143, 208) -- zero for the default "constant", otherwise the feat byte at the remapped
    // coordinate (inside the feat tile for edge / reflect / symmetric; wrap may reach the far side of the frame, which
    // the two-launch path reads from the stage-1 output in HBM)
    auto outside_pixel = [&](int gy, int gx, int c) -> uint32_t {
        if (P.pad_mode == LERF_PAD_CONSTANT) return 0u;
        bool zy, zx;
        const int sy = pad_index(gy, H, P.pad_mode, &zy), sx = pad_index(gx, W, P.pad_mode, &zx);
        const int ry = sy - fy0, rx = sx - fx0;
        if (ry >= 0 && ry < D::FY && rx >= 0 && rx < D::FX) return (uint32_t)Bt[ry * D::FP + rx * CH + c];
        if (FROM_FEAT) return (uint32_t)F.feat[((int64_t)sy * W + sx) * CH + c];
        return 0u;                      // not reached: the host sends wrap padding through the two-launch path
    };

    if (KIND == LERF_KIND_LINEAR) {
        // ---- stage 2, LeRF-L: six byte LUTs (mode x rotation parity), 2 rotations each (eval_lut_sr.py:579-628)
        int8_t* lut = reinterpret_cast<int8_t*>(smem + D::OFF_LUT);
        int16_t* acc = reinterpret_cast<int16_t*>(smem + D::OFF_ACC);
        uint8_t* hq8 = smem + D::OFF_C;                   // input tile is dead: reuse for the u8 hyper values
        const uint8_t* s2 = P.pack + (size_t)(GEN ? P.n1 : 3) * LUT_PAD;
        const int div2 = kQ * 12;
        // every LUT but the first rides in registers behind the lookups of the previous one (as in stage 1), so that only the
        // LDS store of a LUT, not its L2 round trip, stands between two phases
        constexpr int L2N = (LERF_LUT_ENTRIES + 15) / 16, L2TAIL = L2N - 5 * NT;
        static_assert(L2TAIL > 0 && L2TAIL <= NT, "byte LUT = 5 full uint4 rounds + a tail");
        uint4 m0, m1, m2, m3, m4, m5 = make_uint4(0, 0, 0, 0);
#define LERF_L2_LOAD(IDX)                                                                          \
        do {                                                                                       \
            const uint4* s_ = reinterpret_cast<const uint4*>(s2 + (IDX) * LUT_PAD);                \
            m0 = s_[tid]; m1 = s_[tid + NT]; m2 = s_[tid + 2 * NT]; m3 = s_[tid + 3 * NT]; m4 = s_[tid + 4 * NT]; \
            if (tid < L2TAIL) m5 = s_[tid + 5 * NT];                                               \
        } while (0)
#define LERF_L2_STORE()                                                                            \
        do {                                                                                       \
            uint4* d_ = reinterpret_cast<uint4*>(smem + D::OFF_LUT);                               \
            d_[tid] = m0; d_[tid + NT] = m1; d_[tid + 2 * NT] = m2; d_[tid + 3 * NT] = m3; d_[tid + 4 * NT] = m4; \
            if (tid < L2TAIL) d_[tid + 5 * NT] = m5;                                               \
        } while (0)
        // interior tiles of the two-launch path: the seven neighbourhood pixels of a position come from the stage-1 output
        // in HBM/L2 over the vector-memory path, one position ahead (byte_phase_vmem); these phases are LDS-bound like stage 1
#ifndef LERF_S1_LDS_PIXELS
        const bool vm2 = FROM_FEAT && !GEN && interior;
#else
        const bool vm2 = false;
#endif
        const uint8_t* forg = FROM_FEAT ? F.feat + ((int64_t)hy0 * W + hx0) * CH : nullptr;
        const int fpitch = W * CH;
#define LERF_L2(NEXT, MODE, PAR, PH)                                                                       \
        if ((NEXT) < 6) LERF_L2_LOAD(NEXT);                                                                \
        if (vm2) byte_phase_vmem<D::NH, D::HP, MODE, PAR, 2, 2, PH>(lut, forg, fpitch, acc, hq8, div2, 127, tid); \
        else byte_phase<D::NH, D::HP, D::FP, MODE, PAR, 2, 2, PH>(lut, Bt, acc, hq8, hy0, hx0, fy0, fx0, Hc, Wc, div2, 127, tid); \
        __syncthreads();                                                                                   \
        if ((NEXT) < 6) {                                                                                  \
            LERF_L2_STORE();                                                                               \
            __syncthreads();                                                                               \
        }
        copy16<LERF_LUT_ENTRIES>(smem + D::OFF_LUT, s2 + 0 * LUT_PAD, tid);
        __syncthreads();
        if constexpr (GEN) {
            // 2 * n2 LUTs (pattern x rotation parity), offsets of the two rotations of LUT l in P.s2off[l]
            const int nl = 2 * P.n2, dv = kQ * 4 * P.n2;
#pragma unroll 1
            for (int l = 0; l < nl; ++l) {
                const bool more = l + 1 < nl;
                if (more) LERF_L2_LOAD(l + 1);
                if (FROM_FEAT && interior)
                    byte_phase_vmem_rt<D::NH, D::HP, 2>(lut, forg, fpitch, acc, hq8, dv, 127, tid, P.s2dyx[l], l == 0, !more);
                else
                byte_phase_rt<D::NH, D::HP, D::FP, 2>(lut, Bt, acc, hq8, hy0, hx0, fy0, fx0, Hc, Wc, dv, 127, tid, P.s2off[l], l == 0, !more);
                __syncthreads();
                if (more) {
                    LERF_L2_STORE();
                    __syncthreads();
                }
            }
        } else {
        LERF_L2(1, 's', 0, 0)
        LERF_L2(2, 's', 1, 1)
        LERF_L2(3, 'c', 0, 1)
        LERF_L2(4, 'c', 1, 1)
        LERF_L2(5, 't', 0, 1)
        LERF_L2(6, 't', 1, 2)
        }
#undef LERF_L2
#undef LERF_L2_LOAD
#undef LERF_L2_STORE
        // pack (alpha_q, 0, 0, feat-or-0) dwords for stage 3
        uint32_t tmp[(D::NH + NT - 1) / NT];
#pragma unroll
        for (int k = 0; k < (D::NH + NT - 1) / NT; ++k) {
            int p = k * NT + tid;
            tmp[k] = 0;
            if (p < D::NH) {
                bool inside;
                int a = center_addr<D::HP, D::FP>(p, hy0, hx0, fy0, fx0, Hc, Wc, &inside);
                uint32_t px = 0u;
                if (inside) px = (uint32_t)Bt[a];
                else if (P.pad_mode != LERF_PAD_CONSTANT) {
                    const int ry = p / D::HP, r3 = p - ry * D::HP, rx = r3 / CH;
                    px = outside_pixel(hy0 + ry, hx0 + rx, r3 - rx * CH);
                }
                tmp[k] = (uint32_t)hq8[p] | (px << 24);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (D::NH + NT - 1) / NT; ++k) {
            int p = k * NT + tid;
            if (p < D::NH) Dt[p] = tmp[k];
        }
    } else {
        // ---- stage 2, LeRF-G: packed 3-channel LUT pieces, pixels binned by the top-axis level of their centre
        constexpr int MAXR = D::MAXR;
        uint16_t* lst = reinterpret_cast<uint16_t*>(smem + D::OFF_LST);
        int* tab = reinterpret_cast<int*>(smem + D::OFF_TAB);                // [wave][bin] counts
        // Counting sort of the hyper-region positions by bin, every bin padded to whole waves: list entry i belongs to
        // slot round i / NT of thread i % NT, so a 64-entry chunk (one wave in one round) never mixes bins.  One barrier:
        // per-thread histograms -> wave scans -> the 16 x 4 wave totals through LDS -> every wave derives its own bases.
        // Positions of the hyper region that lie outside the frame (tiles on the right / bottom edge, the halo ring of edge
        // tiles) are not looked up at all: stage 3 reads them as replicas of the clamped position (edge-padded hyper maps,
        // zero image), which fill_outside() below copies once the in-frame values exist.  A 28-row strip tile or the
        // last tile row of a 1080-row frame then costs what its in-frame part costs.
        for (int i = tid; i < MAXR * NT / 2; i += NT) reinterpret_cast<uint32_t*>(lst)[i] = 0xFFFFFFFFu;
        constexpr int KH = (D::NH + NT - 1) / NT;
        static_assert(KH <= 15 && NBIN <= 4 && NW == 16, "nibble counts, one byte field per bin, one lane per (wave, bin)");
        uint32_t qlo = 0, qhi = 0;              // 4 bits per owned position p = tid * KH + k: its bin, 15 = not looked up (thread-major lists = position order)
        uint32_t xa, xb;                        // per-thread counts of bins (0, 1) and (2, 3), two 16-bit fields per register
        {
            uint32_t hist = 0;                  // one byte per bin
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int p = tid * KH + k;
                uint32_t q = 15;
                if (p < D::NH) {
                    bool in = true;
                    const int a = center_addr<D::HP, D::FP>(p, hy0, hx0, fy0, fx0, Hc, Wc, &in);
#ifndef LERF_FULL_RING
                    // The last row and the last column of the hyper region are never read by an owned output: a tile owns the
                    // outputs whose first tap lies in [t0 - S/2, t0 + T - S/2), so their taps end at t0 + T + S/2 - 2, one short of
                    // the region's end (the frame's own trailing outputs reach it -- as out-of-frame positions, which fill_outside()
                    // supplies).  Not looked up: 3 % of the positions (66 x 66 -> 65 x 65), 13 slot rounds instead of 14.
                    if (D::HR > 0) {
                        const int ry = p / D::HP, r3 = p - ry * D::HP;
                        if (ry == D::HY - 1 || r3 >= (D::HX - 1) * CH) in = false;
                    }
#endif
                    if (in) q = bin_of_level((uint32_t)Bt[a] >> 4);
                }
                if (k < 8) qlo |= q << (4 * k); else qhi |= q << (4 * (k - 8));
                hist += q < NBIN ? 1u << (8 * q) : 0u;
            }
            xa = (hist & 0xFFu) | ((hist << 8) & 0xFF0000u);
            xb = ((hist >> 16) & 0xFFu) | ((hist >> 8) & 0xFF0000u);
        }
        // wave-inclusive scans of the packed count pairs (row shifts + the two row broadcasts of the DPP unit)
        auto wave_scan = [](uint32_t v) {
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);     // row_shr:1
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);     // row_shr:2
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);     // row_shr:4
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);     // row_shr:8
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31
            return v;
        };
        const uint32_t ia = wave_scan(xa), ib = wave_scan(xb);
        if (lane == 63) {
            int* t = tab + wave * 4;
            t[0] = (int)(ia & 0xFFFFu); t[1] = (int)(ia >> 16); t[2] = (int)(ib & 0xFFFFu); t[3] = (int)(ib >> 16);
        }
        __syncthreads();
        // Every wave turns the 16 x 4 wave counts into what it needs itself (lane = w * 4 + b), in registers: the list base
        // of each of its own bins, and -- the same in all waves -- each bin's first chunk and one past its last (a byte
        // each) and the set of non-empty bins.  No second barrier, no table to read back.
        uint32_t ne_bins, cs_pack, ce_pack;          // wave-uniform
        uint32_t cur01, cur23;                       // this thread's list cursors, two 16-bit fields per register
        {
            const int bb = lane & 3;                             // lane = w * 4 + bb
            const int c = tab[lane];
            int incl = c;                                        // scan over the waves of a bin: lanes 4 apart
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const int total = __shfl(incl, 60 + bb);
            const int padded = (total + 63) & ~63;
            int start = padded;                                  // scan over the bins: 4 neighbouring lanes
#pragma unroll
            for (int d = 1; d < 4; d <<= 1) {
                const int up = __shfl_up(start, d, 4);
                if (bb >= d) start += up;
            }
            start -= padded;
            const int base = start + incl - c;                   // list base of (wave w, bin bb)
            const int wl = __builtin_amdgcn_readfirstlane(wave) * 4;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(base, wl), b1 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 1),
                           b2 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 2), b3 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 3);
            const uint32_t ea = ia - xa, eb = ib - xb;           // counts of the lanes below
            cur01 = ((b0 + (ea & 0xFFFFu)) & 0xFFFFu) | ((b1 + (ea >> 16)) << 16);
            cur23 = ((b2 + (eb & 0xFFFFu)) & 0xFFFFu) | ((b3 + (eb >> 16)) << 16);
            const int sc = start >> 6, ec = (start + padded) >> 6;           // chunks: < 256
            cs_pack = (uint32_t)__builtin_amdgcn_readlane(sc, 0) | ((uint32_t)__builtin_amdgcn_readlane(sc, 1) << 8) |
                      ((uint32_t)__builtin_amdgcn_readlane(sc, 2) << 16) | ((uint32_t)__builtin_amdgcn_readlane(sc, 3) << 24);
            ce_pack = (uint32_t)__builtin_amdgcn_readlane(ec, 0) | ((uint32_t)__builtin_amdgcn_readlane(ec, 1) << 8) |
                      ((uint32_t)__builtin_amdgcn_readlane(ec, 2) << 16) | ((uint32_t)__builtin_amdgcn_readlane(ec, 3) << 24);
            ne_bins = (uint32_t)(__ballot(total > 0) & 0xFull);
        }
        // The next piece rides in registers behind the lookups of the current one: NSLAB x 16 bytes per thread.
        uint4 pr[NSLAB];
#pragma unroll
        for (int i = 0; i < NSLAB; ++i) pr[i] = make_uint4(0, 0, 0, 0);
        const uint8_t* s2p = P.pack + (size_t)(GEN ? P.n1 : 3) * LUT_PAD;          // [LUT l][bin b][PIECE_BYTES], blocks pre-permuted
        const int NL2 = GEN ? 2 * P.n2 : 6;                // stage-2 LUTs: pattern x rotation parity
        auto pre_load = [&](int l, int bq) {
            const uint4* s_ = reinterpret_cast<const uint4*>(s2p + ((size_t)l * NBIN + bq) * PIECE_BYTES) + (wave * 64 + lane);
#pragma unroll
            for (int i = 0; i < NSLAB; ++i) pr[i] = s_[i * NT];
        };
        const int nbins = __builtin_popcount(ne_bins);
        const int nph = nbins * NL2;
        if (nph > 0) pre_load(0, __builtin_ctz(ne_bins));      // in flight during the scatter and the slot set-up
        {
            // scatter: a position's list entry = the thread's cursor of its bin, which then moves on (registers only).
            // The entry is the position's feat-tile ADDRESS (listed positions lie inside the frame, so it is the unclamped
            // one: one increment per position, a row step every HP positions), which the slot set-up then reads back
            // as it is -- no second address computation per slot.
            const int p0 = tid * KH, ry0 = p0 / D::HP;
            uint32_t col = (uint32_t)(p0 - ry0 * D::HP);
            uint32_t ap = (uint32_t)((ry0 + D::HO) * D::FP + D::HO * CH) + col;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const uint32_t q = k < 8 ? (qlo >> (4 * k)) & 0xFu : (qhi >> (4 * (k - 8))) & 0xFu;
                if (q < NBIN) {
                    const uint32_t pair = q < 2 ? cur01 : cur23;
                    const uint32_t v = (q & 1u) ? pair >> 16 : pair & 0xFFFFu;
                    lst[v] = (uint16_t)ap;
                    const uint32_t inc = (q & 1u) ? 0x10000u : 1u;
                    if (q < 2) cur01 += inc; else cur23 += inc;
                }
                ++ap;
                if (++col == (uint32_t)D::HP) { col = 0; ap += (uint32_t)(D::FP - D::HP); }
            }
        }
        __syncthreads();
        // slots -> registers.  Per slot: the feat-tile address of its centre (16 bits, two slots per VGPR) and three
        // accumulators (accA: e0 | e2 << 16 in 16-bit fields; accB: e2 + 256 e1).  Padding lanes of a partly filled wave repeat the wave's
        // first real position (same bin, same LDS words: broadcast reads) into accumulators nobody reads, so the lookup
        // loop needs no per-lane test; `vmask` remembers which slots are real, and the position id comes back out of the
        // address when the sums are finalised (listed positions are inside the frame: the address is not clamped).
        constexpr int MAXP = (MAXR + 1) / 2;
        uint32_t slot2[MAXP];
        uint32_t accA[MAXR], accB[MAXR];
        uint32_t wrounds = 0;                 // bit k: this wave has a real position in round k (wave-uniform)
        uint32_t vmask = 0;                   // bit k: this lane's slot k is real
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const uint32_t p = lst[k * NT + tid];
            uint32_t a = p;
            if (p != 0xFFFFu) vmask |= 1u << k;
            const unsigned long long real = __ballot(p != 0xFFFFu);
            if (real != 0ull) {
                const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)__builtin_ctzll(real));
                if (p == 0xFFFFu) a = a1;
                wrounds |= 1u << k;
            }
            if (k & 1) slot2[k >> 1] |= a << 16; else slot2[k >> 1] = a;
            accA[k] = 0;
            accB[k] = 0;
        }
        __syncthreads();                      // the lists are dead: the first piece may land on them

        LERF_STAMP(7);
        // The piece travels as NSLAB x 16 bytes per thread (wave w, slab i: 1-KiB block B = 16 i + w of the piece, lane L
        // its bytes 16 L..16 L+15) and is stored with ds_write_addtid_b32 (address = M0 + offset + 4*lane, no address
        // VGPR: 128 B/clk/CU against 79 for ds_write_b128).  addtid puts component c of all lanes at block + 256 c + 4 L,
        // so the pack holds every block pre-permuted (global dword 4L+c = logical dword 64c+L) and LDS ends up in
        // natural order.  M0 = the wave's block column; the 16-bit offset reaches 4 slabs.
#define LERF_ADDTID(V, OFF) asm volatile("ds_write_addtid_b32 %0 offset:" #OFF :: "v"(V) : "memory")
#define LERF_ADDTID4(R, OFF0, OFF1, OFF2, OFF3) \
        LERF_ADDTID(R.x, OFF0); LERF_ADDTID(R.y, OFF1); LERF_ADDTID(R.z, OFF2); LERF_ADDTID(R.w, OFF3)
#define LERF_SET_M0(V) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(V) : "memory")
        auto pre_store = [&]() {
            const uint32_t m0a = __builtin_amdgcn_readfirstlane((uint32_t)D::OFF_X + (uint32_t)wave * 1024u);
            // slab i holds blocks 16 i + wave: the last slab is partial (blocks beyond PIECE_BLOCKS carry no data)
#define LERF_SLAB_OK(I) ((I) < NSLAB && (16 * (I) + 15 < PIECE_BLOCKS || 16 * (I) + wave < PIECE_BLOCKS))
            LERF_SET_M0(m0a);
            if (LERF_SLAB_OK(0)) { LERF_ADDTID4(pr[0], 0, 256, 512, 768); }
            if (LERF_SLAB_OK(1)) { LERF_ADDTID4(pr[1 < NSLAB ? 1 : 0], 16384, 16640, 16896, 17152); }
            if (LERF_SLAB_OK(2)) { LERF_ADDTID4(pr[2 < NSLAB ? 2 : 0], 32768, 33024, 33280, 33536); }
            if (LERF_SLAB_OK(3)) { LERF_ADDTID4(pr[3 < NSLAB ? 3 : 0], 49152, 49408, 49664, 49920); }
            if (NSLAB > 4) {
                LERF_SET_M0(m0a + 65536u);
                if (LERF_SLAB_OK(4)) { LERF_ADDTID4(pr[4 < NSLAB ? 4 : 0], 0, 256, 512, 768); }
                if (LERF_SLAB_OK(5)) { LERF_ADDTID4(pr[5 < NSLAB ? 5 : 0], 16384, 16640, 16896, 17152); }
                if (LERF_SLAB_OK(6)) { LERF_ADDTID4(pr[6 < NSLAB ? 6 : 0], 32768, 33024, 33280, 33536); }
                if (LERF_SLAB_OK(7)) { LERF_ADDTID4(pr[7 < NSLAB ? 7 : 0], 49152, 49408, 49664, 49920); }
            }
            if (NSLAB > 8) {
                LERF_SET_M0(m0a + 81920u);       // slab 5's base: M0 stays below 128 KiB, the offset field supplies the rest
                if (LERF_SLAB_OK(8)) { LERF_ADDTID4(pr[8 < NSLAB ? 8 : 0], 49152, 49408, 49664, 49920); }
            }
            static_assert(NSLAB <= 9, "three M0 windows");
#undef LERF_SLAB_OK
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // phases = (non-empty bin) x (6 LUTs).  The next piece is fetched into registers while the current one is being
        // used, so the L2 latency of the piece copies hides behind the lookups.
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // per-bin scalars, refreshed when a bin's first LUT comes up (kept loop-carried on purpose: loop-invariant code
        // motion out of an inner per-LUT loop would park every round's slot address in a register of its own)
        int bq = 0, bq_next = 0, l = 0, bi = 0;
        uint32_t ne_left = ne_bins;               // non-empty bins not yet started
        uint32_t act = 0, qbase = 0;
        for (int ph = 0; ph < nph; ++ph) {
            if (l == 0) {
                // its level range, and this wave's rounds: chunk 16 k + wave inside [cs, ce) -- one scalar bit test per
                // unrolled round
                bq = __builtin_ctz(ne_left);
                ne_left &= ne_left - 1u;
                bq_next = ne_left != 0u ? __builtin_ctz(ne_left) : 0;
                const int cs = (int)((cs_pack >> (8 * bq)) & 0xFFu), ce = (int)((ce_pack >> (8 * bq)) & 0xFFu);
                const int klo = cs > wv ? (cs - wv + 15) >> 4 : 0, khi = ce > wv ? (ce - wv + 15) >> 4 : 0;
                act = wrounds & ((1u << khi) - 1u) & ~((1u << klo) - 1u);
                // LDS address of the piece's logical entry 0 (the piece starts at top-axis level bin_lo(bq))
                qbase = lds_addr(smem + D::OFF_X) - (uint32_t)bin_lo(bq) * (kStrideA * 4u);
            }
            const unsigned long long t_copy = LERF_NOW();
            (void)t_copy;
            pre_store();
            Off3 o0, o1;                                     // LUT l = mode (l>>1), rotation parity (l&1)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                o0.o[i] = P.s2off[l][i];
                o1.o[i] = P.s2off[l][3 + i];
            }
            const uint32_t bt_a = lds_addr(Bt);
            const unsigned st_a = kStrideA * 4, st_b = kStrideB * 4, st_c = kStrideC * 4, st_d = kStrideD * 4;   // axis strides, bytes
            __syncthreads();
            if (l < NL2 - 1) pre_load(l + 1, bq);
            else if (bi + 1 < nbins) pre_load(0, bq_next);
            LERF_STAMP_ADD(8, t_copy);
            const unsigned long long t_look = LERF_NOW();
            (void)t_look;
#pragma unroll
            for (int k = 0; k < MAXR; ++k) {
                if ((act >> k) & 1u) {
                    const uint32_t sa = (k & 1) ? (slot2[k >> 1] >> 16) : (slot2[k >> 1] & 0xFFFFu);
                    // stage A: the 7 pixel reads of the two rotations (high-half loads, see lds_pixel_hi)
                    const uint32_t cpa = bt_a + sa;
                    uint32_t ra = lds_pixel_hi(cpa);
                    uint32_t rb0 = lds_pixel_hi(cpa + (uint32_t)o0.o[0]), rc0 = lds_pixel_hi(cpa + (uint32_t)o0.o[1]),
                             rd0 = lds_pixel_hi(cpa + (uint32_t)o0.o[2]);
                    uint32_t rb1 = lds_pixel_hi(cpa + (uint32_t)o1.o[0]), rc1 = lds_pixel_hi(cpa + (uint32_t)o1.o[1]),
                             rd1 = lds_pixel_hi(cpa + (uint32_t)o1.o[2]);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb0), "+v"(rc0), "+v"(rd0), "+v"(rb1), "+v"(rc1), "+v"(rd1));
                    // stage B: both walks (LDS addresses into the piece)
                    const int basea = (int)(__umul24(msb_of(ra), kStrideA * 4) + qbase);
                    const unsigned ka = key_of(ra, st_a);
                    const Walk<4> W0 = simplex_walk<4>(ka, basea, rb0, rc0, rd0, st_b, st_c, st_d);
                    const Walk<4> W1 = simplex_walk<4>(ka, basea, rb1, rc1, rd1, st_b, st_c, st_d);
                    // stage C: ten dword gathers in flight together
                    uint32_t d0[5], d1[5];
#pragma unroll
                    for (int n = 0; n < 5; ++n) d0[n] = W0.ld32(n);
#pragma unroll
                    for (int n = 0; n < 5; ++n) d1[n] = W1.ld32(n);
                    __builtin_amdgcn_sched_barrier(0);
                    // stage D: two multiply-adds per corner.  The entry is e0 | 0 << 8 | e2 << 16 | e1 << 24: a 24-bit multiply reads
                    // bits 0..23 only, so the (e0, e2) pair needs no mask; the second one multiplies the entry's HIGH HALF
                    // (e2 + 256 e1, v_mad_u32_u16 with op_sel) without a shift -- accB = sum w e2 + 256 sum w e1, and the e2
                    // sum is already in accA's upper field, so the finalisation takes it out again (round 3: 3 -> 2
                    // instructions per corner, 87 -> 77 per slot)
                    const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
                    const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
                    uint32_t a = accA[k], bb = accB[k];
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        a += __umul24(w0[n], d0[n]);
                        bb = mad_hi16(w0[n], d0[n], bb);
                    }
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        a += __umul24(w1[n], d1[n]);
                        bb = mad_hi16(w1[n], d1[n], bb);
                    }
                    accA[k] = a;
                    accB[k] = bb;
                }
            }
            __syncthreads();
            LERF_STAMP_ADD(9, t_look);
            if (++l == NL2) { l = 0; ++bi; }
        }
        LERF_STAMP(10);
        // finalise: hq = rne(clip(N/192 + 127)); entries are biased by +128 -> 12 lookups * 16 * 128 = 24576
        //           N + 127*192 = field - 24576 + 24384 = field - 192
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if ((vmask >> k) & 1u) {
                const int div2 = GEN ? kQ * 4 * P.n2 : kQ * 12;     // general kernels: 4 rotations x n2 patterns (same bias algebra)
                int n0 = (int)(accA[k] & 0xFFFFu) - div2;
                int n2 = (int)(accA[k] >> 16) - div2;
                int n1 = (int)((accB[k] - (accA[k] >> 16)) >> 8) - div2;      // accB = (e2 sum) + 256 (e1 sum), see the slot's MACs
                uint32_t h0 = (uint32_t)rne_div_clip255_fast(n0, div2);
                uint32_t h1 = (uint32_t)rne_div_clip255_fast(n1, div2);
                uint32_t h2 = (uint32_t)rne_div_clip255_fast(n2, div2);
                // feat-tile address -> hyper-region position: row (ry + R2) of pitch FP -> row ry of pitch HP, R2 pixels left
                const uint32_t a = (k & 1) ? (slot2[k >> 1] >> 16) : (slot2[k >> 1] & 0xFFFFu);
                const uint32_t row = a / (uint32_t)D::FP;
                const uint32_t p = a - row * (uint32_t)(D::FP - D::HP) - (uint32_t)(D::HO * D::HP + D::HO * CH);
                Dt[p] = h0 | (h1 << 8) | (h2 << 16) | ((uint32_t)Bt[a] << 24);
            }
        }
    }

    if (!EMIT && !WARP && KIND == LERF_KIND_GAUSS && Hc >= 0) {
        // fill_outside: out-of-frame positions of the hyper region = the clamped position's hyper bytes, image byte 0
        __syncthreads();
        for (int p = tid; p < D::NH; p += NT) {
            const int ry = p / D::HP, r3 = p - ry * D::HP, rx = r3 / CH, c = r3 - rx * CH;
            const int gy = hy0 + ry, gx = hx0 + rx;
            const int cy = clampi(gy, 0, H - 1), cx = clampi(gx, 0, W - 1);
            if (cy != gy || cx != gx)
                Dt[p] = (Dt[(cy - hy0) * D::HP + (cx - hx0) * CH + c] & 0x00FFFFFFu) | (outside_pixel(gy, gx, c) << 24);
        }
    }

    if constexpr (WARP) {
        // ---- stage 3 of the warp: the output pixels this tile owns, taps from the tile's packed dwords in LDS
        __syncthreads();
        warp_tile<KIND, D::HP>(P, Dt, hy0, hx0, ty0, tx0, H, W, tyi * F.tiles_x + txi, outp, tid);
        return;
    }
    if (EMIT) {
        __syncthreads();
        uint32_t* eo = F.emit;
        const int rows = min(TH, H - ty0), cols3 = min(TW, W - tx0) * CH;
        for (int il = wave; il < rows; il += NW) {
            const uint32_t* srow = Dt + (il + D::HR) * D::HP + D::HR * CH;
            uint32_t* drow = eo + ((int64_t)(ty0 + il) * W + tx0) * CH;
            for (int x = lane; x < cols3; x += 64) drow[x] = srow[x];
        }
        return;
    }

    // ---- stage 3 geometry of the owned output block (staged here for S = 4; S = 2 did it at kernel start)
    if (!D::GEO_EARLY) {
        geo_search();
        __syncthreads();
        geo_stage();
    }
    __syncthreads();
    const int i0 = ctl[16], i1 = ctl[17], j0 = ctl[18], j1 = ctl[19];
    const int nrow = i1 - i0, ncol = j1 - j0;

    LERF_STAMP(11);
    // ---- stage 3.  Consecutive output rows that start at the same source row (2 rows at x2, 3 at x3 ...) form a row
    //      group and share their taps: one task = (row group, one 4-byte-aligned output dword column); the tap loads, the
    //      u8 -> f32 conversions and the column-only terms are done once per group, the rows pay one multiply and two
    //      FMAs per tap.  Same code for the 2x2 and the 4x4 support.
    {
        constexpr int SS = S * S;
        constexpr int GMAX = S == 2 ? 5 : 3;                      // rows per group (larger runs are split)
        int* g_grp = reinterpret_cast<int*>(g_dc + D::GEO_COLS * S);
        const int ncolc = ncol * CH;
        // bytes per output row: dense, or the caller's pitch (a rank's block of 1919 / 1921 output columns lives in rows padded to a
        // 16-byte multiple so that its tiles keep the dword-column / block tasks -- dense, every row was a group of its own)
        const int64_t rowpitch = P.out_pitch > 0 ? (int64_t)P.out_pitch : (int64_t)F.oW * CH;      // (ragged launches: always dense)
        uint8_t* seg0 = outp + (int64_t)i0 * rowpitch + (int64_t)j0 * CH;
        const bool rows_align = (rowpitch & 3) == 0;              // dword columns line up across the rows of a group
        // dword columns per row: when every row of the block starts on a 4-byte boundary (a0 == 0 below) the block needs
        // no slack column, and a 384-byte tile row is exactly 96 dwords = whole 128-byte lines per wave store
        const bool seg_aligned = rows_align && (reinterpret_cast<uintptr_t>(seg0) & 3u) == 0;
        const int ndw = seg_aligned ? (ncolc + 3) >> 2 : (ncolc + 6) >> 2;
        if (wave == 0) {
            int ng = 0;
            for (int base = 0; base < nrow; base += 64) {
                const int il = base + lane;
                bool start = false;
                if (il < nrow)
                    start = il == 0 || !rows_align || g_lr[il] != g_lr[il - 1] || (il >= GMAX && g_lr[il - GMAX] == g_lr[il]);
                const unsigned long long m = __ballot(start);
                if (start) g_grp[ng + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = il;
                ng += __popcll(m);
            }
            if (lane == 0) g_grp[ng] = nrow;
            // every group of the tile the same size (integer scale factors, away from the frame's top and bottom): the
            // task loop then runs with that size as a compile-time constant, without a row test per tap set
            const int g0 = ng > 0 ? g_grp[1] - g_grp[0] : 0;
            bool same = true;
            for (int base = 0; base < ng; base += 64) {
                const int idx = base + lane;
                if (idx < ng && g_grp[idx + 1] - g_grp[idx] != g0) same = false;
            }
            const bool uniform = __ballot(!same) == 0ull;
            if (lane == 0) {
                ctl[21] = ng;
                ctl[22] = uniform ? g0 : 0;
                ctl[23] = 0;
            }
            if constexpr (KIND == LERF_KIND_LINEAR && S == 2 && D::OUT_FITS) {
                bool small = true;                               // no group of more than two rows (block tasks of LeRF-L)
                for (int base = 0; base < ng; base += 64) {
                    const int idx = base + lane;
                    if (idx < ng && g_grp[idx + 1] - g_grp[idx] > 2) small = false;
                }
                const bool ok = __ballot(!small) == 0ull;
                if (lane == 0) ctl[26] = ok ? 1 : 0;
            }
        } else if (wave == 1 && KIND == LERF_KIND_LINEAR && S == 2 && D::OUT_FITS) {
            // LeRF-L block tasks: the columns in groups of at most two that share their taps (runs of equal left taps, every
            // second column of a run starts a group); the table lies behind the output image, over the dead stage-2 areas
            int* g_cgrp = reinterpret_cast<int*>(smem + D::OFF_CGRP);
            constexpr int NCMAX = 2 * TW + 14;
            int nc = 0;
            for (int base = 0; base < ncol; base += 64) {
                const int jl = base + lane;
                bool start = false;
                if (jl < ncol) {
                    int back = 0;                                // columns of the same run to the left
                    while (back < 16 && jl - back - 1 >= 0 && g_lc[jl - back - 1] == g_lc[jl]) ++back;
                    start = (back & 1) == 0;
                }
                const unsigned long long m = __ballot(start);
                if (start) {
                    const int idx = nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (idx < NCMAX) g_cgrp[idx] = jl;
                }
                nc += __popcll(m);
            }
            if (lane == 0) {
                if (nc <= NCMAX) g_cgrp[nc] = ncol;
                ctl[25] = nc;
                ctl[27] = (nc <= NCMAX) ? 1 : 0;
            }
        } else if (wave == 1) {
            // the columns (block tasks below): do they pair up as (2h, 2h + 1) -- code 1 -- or, behind a leading single (the
            // frame's left edge), as (2h - 1, 2h) -- code 2?  A trailing single is fine either way.
            bool ok0 = true, ok1 = true, step1 = true;
            for (int base = 0; base < ncol; base += 64) {
                const int jl = base + lane;
                if (jl + 1 < ncol && g_lc[jl + 1] != g_lc[jl]) { if (jl & 1) ok1 = false; else ok0 = false; }
                if (!(jl & 1) && jl + 2 < ncol && g_lc[jl + 2] != g_lc[jl] + 1) step1 = false;
            }
            const bool p0 = __ballot(!ok0) == 0ull, p1 = __ballot(!ok1) == 0ull, q1 = __ballot(!step1) == 0ull;
            if (lane == 0) { ctl[24] = p0 ? 1 : (p1 ? 2 : 0); ctl[28] = q1 ? 1 : 0; }   // [28]: pair h + 1 starts one tap right of pair h (x2)
        } else if (wave == 2 && KIND == LERF_KIND_GAUSS) {
            // the rows, the same question on the rows' first taps
            bool ok0 = true, ok1 = true, step1 = true;
            for (int base = 0; base < nrow; base += 64) {
                const int il = base + lane;
                if (il + 1 < nrow && g_lr[il + 1] != g_lr[il]) { if (il & 1) ok1 = false; else ok0 = false; }
                if (!(il & 1) && il + 2 < nrow && g_lr[il + 2] != g_lr[il] + 1) step1 = false;
            }
            const bool p0 = __ballot(!ok0) == 0ull, p1 = __ballot(!ok1) == 0ull, q1 = __ballot(!step1) == 0ull;
            if (lane == 0) { ctl[26] = p0 ? 1 : (p1 ? 2 : 0); ctl[29] = q1 ? 1 : 0; }
        }
        __syncthreads();
        const int ngrp = ctl[21];
        const int gsame = __builtin_amdgcn_readfirstlane(ctl[22]);
        const unsigned magic = (unsigned)((0x100000000ull + (unsigned)ndw - 1) / (unsigned)(ndw > 0 ? ndw : 1));
        const float ms255 = P.max_sigma * (1.0f / 255.0f);
        const int ntask = ngrp * ndw;
        uint32_t* tq = reinterpret_cast<uint32_t*>(smem + D::OFF_TQ);
        int* tq_count = ctl + 23;                               // zeroed with the group table below
        auto run_tasks = [&](auto gs_const) {
        constexpr int GS = decltype(gs_const)::value;          // rows per group, 0 = read it per group
        constexpr int GN = GS > 0 ? GS : GMAX;
        for (int t = tid; t < ntask; t += NT) {
            const int g = ndw == 1 ? t : (int)__umulhi((unsigned)t, magic);      // (ceil(2^32 / 1) does not fit the 32-bit magic)
            const int dw = t - g * ndw;
            const int il0 = g_grp[g];
            const int gs = GS > 0 ? GS : g_grp[g + 1] - il0;
            uint8_t* seg = seg0 + il0 * rowpitch;
            const int a0 = (int)(reinterpret_cast<uintptr_t>(seg) & 3u);
            const int b0 = dw * 4 - a0;
            const int lr = g_lr[il0];
            uint32_t packed[GN];
            // tie detection: the 2x2 kernels keep every output's distance from its rounded value and a running maximum
            // (one v_max per output, the ties are looked for only when the maximum says there is one); the 4x4 kernel has
            // no registers for that and sets a bit per output
            constexpr bool DIST = S == 2;
            float dist[DIST ? GN * 4 : 1];                        // [r*4+u]
            float dmax = 0.0f;
            unsigned tiebits = 0;                                 // bit r*4+u
#pragma unroll
            for (int r = 0; r < GN; ++r) packed[r] = 0;
            s3::f2 DXP[S];                                        // packed 4x4 path: (row 0, row 1) distances of row tap b
            if constexpr (KIND == LERF_KIND_GAUSS && S == 4 && GS == 2) {
#pragma unroll
                for (int b = 0; b < S; ++b) { DXP[b].x = g_dr[il0 * S + b]; DXP[b].y = g_dr[(il0 + 1) * S + b]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int xc = min(max(b0 + u, 0), ncolc - 1);    // clamped; invalid bytes are not stored
                const int jl = xc / CH;
                const int c = xc - jl * CH;
                const int lc = g_lc[jl];
                if constexpr (KIND == LERF_KIND_GAUSS && S == 4 && GS == 2) {
                    // The 4x4 support, two rows per group, in PACKED float32 over the two rows (lane x = row 0, lane y = row 1):
                    // per tap and row PAIR one v_pk_mul (tx), two v_pk_fma (the form), one v_pk_add and one v_pk_fma (the sums)
                    // instead of ten scalar operations; a tap's four scalars sit two to a register pair -- (p0, ty^2) and
                    // (k1, value) -- and are read through op_sel.  Same operations in the same order: the same bytes.
                    s3::f2 PT[SS], KV[SS];
#pragma unroll
                    for (int a = 0; a < S; ++a) {
                        const float dy = g_dc[jl * S + a];
#pragma unroll
                        for (int b = 0; b < S; ++b) {
                            const uint32_t d = Dt[(lr + b) * D::HP + (lc + a) * CH + c];
                            const float tyv = s3::gauss_t_u8((float)((d >> 16) & 0xFFu), dy);
                            PT[a * S + b].x = s3::gauss_m2rho_u8((float)(d & 0xFFu)) * tyv;
                            PT[a * S + b].y = tyv * tyv;
                            KV[a * S + b].x = (float)((d >> 8) & 0xFFu);
                            KV[a * S + b].y = (float)(d >> 24);
                        }
                    }
                    s3::f2 NUM, DEN;
#pragma unroll
                    for (int a = 0; a < S; ++a)
#pragma unroll
                        for (int b = 0; b < S; ++b) {
                            const int k = a * S + b;
                            const s3::f2 TX = s3::pk_mul_blo(DXP[b], KV[k]);                       // k1 * dx[r][b]
                            const s3::f2 E = s3::pk_fma_pb<false>(TX, PT[k], s3::pk_fma_ppb<true>(TX, PT[k]));   // fma(tx, p0, fma(tx, tx, ty2))
                            s3::f2 W;
                            W.x = __builtin_amdgcn_exp2f(-E.x);
                            W.y = __builtin_amdgcn_exp2f(-E.y);
                            if (k == 0) {
                                DEN = W;
                                NUM = s3::pk_mul_bhi_after_trans(W, KV[k]);
                            } else {
                                DEN = DEN + W;                                                  // (compiler-generated: see the block tasks)
                                NUM = s3::pk_fma_bhi_after_trans(W, KV[k], NUM);
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float xf = s3::finish_div(r ? NUM.y : NUM.x, r ? DEN.y : DEN.x);
                        bool tie;
                        packed[r] = s3::pack_u8_tie(xf, u, packed[r], &tie);
                        if (tie) tiebits |= 1u << (r * 4 + u);
                    }
                    continue;
                }
                // shared by the rows of the group: per tap (a = column offset major, numpy meshgrid 'xy' :95-98; b = row offset)
                float p0[SS], k1[SS], ty[SS], v[SS];
#pragma unroll
                for (int a = 0; a < S; ++a) {
                    const float dy = g_dc[jl * S + a];
#pragma unroll
                    for (int b = 0; b < S; ++b) {
                        const uint32_t d = Dt[(lr + b) * D::HP + (lc + a) * CH + c];
                        v[a * S + b] = (float)(d >> 24);
                        const float k0 = (float)(d & 0xFFu);
                        if (KIND == LERF_KIND_GAUSS) {
                            // column-only terms of the quadratic form: p0 <- (-2 rho) ty, ty <- ty^2 (gauss_form_cols)
                            const float tyv = s3::gauss_t_u8((float)((d >> 16) & 0xFFu), dy);
                            p0[a * S + b] = s3::gauss_m2rho_u8(k0) * tyv;
                            k1[a * S + b] = (float)((d >> 8) & 0xFFu);
                            ty[a * S + b] = tyv * tyv;
                        } else {
                            const float alpha = s3::lin_alpha_u8(k0, ms255);
                            p0[a * S + b] = alpha;
                            ty[a * S + b] = s3::lin_factor(alpha, dy, s3::dist_class_f(dy));
                            k1[a * S + b] = 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < GN; ++r) {
                    if (r < gs) {
                        float e[SS];
#pragma unroll
                        for (int a = 0; a < S; ++a)
#pragma unroll
                            for (int b = 0; b < S; ++b) {
                                const float dx = g_dr[(il0 + r) * S + b];
                                if (KIND == LERF_KIND_GAUSS)
                                    e[a * S + b] = s3::gauss_form_cols(s3::gauss_t_u8(k1[a * S + b], dx), ty[a * S + b], p0[a * S + b]);
                                else
                                    e[a * S + b] = s3::lin_factor(p0[a * S + b], dx, s3::dist_class_f(dx)) * ty[a * S + b];
                            }
                        const float xf = s3::finish<KIND == LERF_KIND_GAUSS, SS, true, true, true>(e, v);
                        if (DIST) {
                            packed[r] = s3::pack_u8_dist(xf, u, packed[r], &dist[DIST ? r * 4 + u : 0]);
                            dmax = __builtin_fmaxf(dmax, __builtin_fabsf(dist[DIST ? r * 4 + u : 0]));
                        } else {
                            bool tie;
                            packed[r] = s3::pack_u8_tie(xf, u, packed[r], &tie);
                            if (tie) tiebits |= 1u << (r * 4 + u);
                        }
                    }
                }
            }
            if ((DIST ? dmax > 0.5f - s3::kTieEps : tiebits != 0) && F.dis_r64 != nullptr) {
                unsigned tiemask = tiebits;
                if (DIST) {
#pragma unroll
                    for (int q = 0; q < GN * 4; ++q)
                        if ((GS > 0 || (q >> 2) < gs) && __builtin_fabsf(dist[DIST ? q : 0]) > 0.5f - s3::kTieEps) tiemask |= 1u << q;
                }
                // rare: the output is re-evaluated in float64 exactly as the reference does (lerf_stage3.h, tie guard)
                while (tiemask != 0) {                                     // one round per queued output of the lane, not per output
                    const int q = __builtin_ctz(tiemask);
                    tiemask &= tiemask - 1;
                    const int r = q >> 2, u = q & 3;
                    const int xc = min(max(b0 + u, 0), ncolc - 1);
                    // queued for the pass behind the task loop; only a full queue is worked off on the spot
                    const int slot = atomicAdd(tq_count, 1);
                    if (slot < P.tq_cap) {
                        tq[slot] = ((uint32_t)(il0 + r) << 16) | (uint32_t)xc;
                        continue;
                    }
                    const int jl = xc / CH;
                    const int c = xc - jl * CH;
                    const int lc = g_lc[jl];
                    uint32_t dd[SS];
                    double dx64[S], dy64[S];
#pragma unroll
                    for (int b = 0; b < S; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il0 + r) * S + b];
#pragma unroll
                    for (int a = 0; a < S; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl) * S + a];
#pragma unroll
                    for (int a = 0; a < S; ++a)
#pragma unroll
                        for (int b = 0; b < S; ++b) dd[a * S + b] = Dt[(lr + b) * D::HP + (lc + a) * CH + c];
                    const uint32_t r8 = s3::resolve_u8<KIND == LERF_KIND_GAUSS, S>(dd, dx64, dy64, P.max_sigma);
#pragma unroll
                    for (int rr = 0; rr < GN; ++rr)
                        if (rr == r) packed[rr] = (packed[rr] & ~(0xFFu << (8 * u))) | (r8 << (8 * u));
                }
            }
#pragma unroll
            for (int r = 0; r < GN; ++r) {
                if (r < gs) {
                    uint8_t* sr = seg + r * rowpitch;
                    if (b0 >= 0 && b0 + 3 < ncolc) {
                        // streaming store: the output is never re-read here, keep the LUT pack resident in L2 instead
                        __builtin_nontemporal_store(packed[r], reinterpret_cast<uint32_t*>(sr + b0));
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (b0 + u >= 0 && b0 + u < ncolc) sr[b0 + u] = (uint8_t)(packed[r] >> (8 * u));
                    }
                }
            }
        }
        };
        // ---- block tasks: tiles whose output rows AND columns come in pairs that share their taps (integer x2, away from
        //      the frame's top and left).  One task = the 2 x 2 outputs of a (row pair, column pair), all CH channels: the
        //      four tap dwords of a channel are loaded and converted once for four outputs instead of once for two, the
        //      column-only terms serve both rows and the row-only products both columns (43 instead of 63 VALU instructions
        //      per output), and a tile is exactly 4 tasks per thread.  The bytes are assembled in an LDS image of the tile's
        //      output block (ties are patched there) and leave as whole 16-byte chunks.  Same arithmetic, operation for
        //      operation, as run_tasks: the outputs are bit-identical.
        constexpr bool BLK = KIND == LERF_KIND_GAUSS && (S == 2 || S == 4) && D::OUT_FITS;      // (round 4: the 4 x 4 support too)
        uint8_t* outt = smem + D::OFF_OUT;
        const int ophase = (int)(reinterpret_cast<uintptr_t>(seg0) & 15u);
#ifndef LERF_NO_BLOCK_TASKS
        const int rcode = __builtin_amdgcn_readfirstlane(ctl[26]), ccode = __builtin_amdgcn_readfirstlane(ctl[24]);
        const bool blk = BLK && rows_align && rcode != 0 && ccode != 0 && (rowpitch & 15) == 0 && nrow <= D::OUT_ROWS &&
                         ophase + ncolc <= D::OUT_PITCH;
        // interior tiles: pairs from row 0 / column 0 on, even counts -- no validity tests; tiles at the frame's edges: a leading
        // and / or trailing single row or column, handled as a pair whose other member is not stored
        const bool blk_uni = blk && rcode == 1 && ccode == 1 && ((nrow | ncol) & 1) == 0;
#else
        const bool blk = false, blk_uni = false;
#endif
        auto run_blocks = [&](auto uni_const) {
            constexpr bool UNI = decltype(uni_const)::value;
            if constexpr (BLK) {
            constexpr int SS2 = S * S;
            const int rofs = UNI ? 0 : rcode - 1, cofs = UNI ? 0 : ccode - 1;
            const int ncg = (ncol + cofs + 1) >> 1;
            const int nblk = ((nrow + rofs + 1) >> 1) * ncg;
            const unsigned magicc = (unsigned)((0x100000000ull + (unsigned)ncg - 1) / (unsigned)(ncg > 0 ? ncg : 1));
            for (int t = tid; t < nblk; t += NT) {
                const int g = ncg == 1 ? t : (int)__umulhi((unsigned)t, magicc);
                const int h = t - g * ncg;
                const int il0 = 2 * g - rofs, jl0 = 2 * h - cofs;
                const int lr = g_lr[UNI ? il0 : max(il0, 0)], lc = g_lc[UNI ? jl0 : max(jl0, 0)];
                // distances as PAIRS: DX[b] = (row 0, row 1) of row tap b, DY[a] = (column 0, column 1) of column tap a
                s3::f2 DX[S], DY[S];
#pragma unroll
                for (int b = 0; b < S; ++b) { DX[b].x = g_dr[il0 * S + b]; DX[b].y = g_dr[il0 * S + S + b]; }
#pragma unroll
                for (int a = 0; a < S; ++a) { DY[a].x = g_dc[jl0 * S + a]; DY[a].y = g_dc[jl0 * S + S + a]; }
                const uint32_t* dp = Dt + lr * D::HP + lc * CH;
                uint8_t* ob = outt + il0 * D::OUT_PITCH + ophase + jl0 * CH;
                float dist[4 * CH];                                  // [c][r][q]
                float dmax = 0.0f;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    // packed over the two COLUMNS q of the block (lane x = column 0, lane y = column 1); the scalars of a tap
                    // (its value, m2rho, k1, k2) sit in the low halves of register pairs and are read through op_sel
                    s3::f2 NUM[2], DEN[2];                           // [row r]
#pragma unroll
                    for (int a = 0; a < S; ++a)
#pragma unroll
                        for (int b = 0; b < S; ++b) {
                            const uint32_t d = dp[b * D::HP + a * CH + c];
                            s3::f2 V, K1, K2, M2;
                            V.x = V.y = (float)(d >> 24);                          // (both lanes: the weights' consumers below are plain vector code)
                            K1.x = (float)((d >> 8) & 0xFFu);
                            M2.x = s3::gauss_m2rho_u8((float)(d & 0xFFu));
                            K2.x = (float)((d >> 16) & 0xFFu);
                            const s3::f2 TYV = s3::pk_mul_blo(DY[a], K2);          // k2 * dy[q][a]
                            const s3::f2 P0 = s3::pk_mul_blo(TYV, M2);             // m2rho * tyv
                            const s3::f2 TY2 = s3::pk_mul(TYV, TYV);
                            const s3::f2 TX = s3::pk_mul_blo(DX[b], K1);           // lanes = ROWS here: k1 * dx[r][b]
                            // e[r][q] = fma(tx_r, p0_q, fma(tx_r, tx_r, ty2_q)), w = exp2(-e), num += w v, den += w
                            const s3::f2 E0 = s3::pk_fma_ab<false>(TX, P0, s3::pk_fma_aa<false>(TX, TY2));
                            const s3::f2 E1 = s3::pk_fma_ab<true>(TX, P0, s3::pk_fma_aa<true>(TX, TY2));
                            s3::f2 W0, W1;
                            W0.x = __builtin_amdgcn_exp2f(-E0.x); W0.y = __builtin_amdgcn_exp2f(-E0.y);
                            W1.x = __builtin_amdgcn_exp2f(-E1.x); W1.y = __builtin_amdgcn_exp2f(-E1.y);
                            // The consumers of the v_exp results are COMPILER-generated packed operations, not inline assembly: gfx950
                            // needs a wait state between a transcendental instruction and a VALU read of its result, which the
                            // compiler inserts for its own instructions only (an inline-asm consumer right behind v_exp read the stale
                            // register: nondeterministic bytes).
                            if (a == 0 && b == 0) {
                                DEN[0] = W0; DEN[1] = W1;
                                NUM[0] = W0 * V; NUM[1] = W1 * V;
                            } else {
                                NUM[0] = __builtin_elementwise_fma(W0, V, NUM[0]); NUM[1] = __builtin_elementwise_fma(W1, V, NUM[1]);
                                DEN[0] = DEN[0] + W0; DEN[1] = DEN[1] + W1;
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float den = q ? DEN[r].y : DEN[r].x, num = q ? NUM[r].y : NUM[r].x;
                            const float xf = s3::finish_div(num, den);
                            const float rr = __builtin_rintf(xf);
                            const float ds = xf - rr;
                            const bool have = UNI || ((unsigned)(il0 + r) < (unsigned)nrow && (unsigned)(jl0 + q) < (unsigned)ncol);
                            dist[(c * 2 + r) * 2 + q] = have ? ds : 0.0f;
                            dmax = __builtin_fmaxf(dmax, __builtin_fabsf(dist[(c * 2 + r) * 2 + q]));
                            if (have) ob[r * D::OUT_PITCH + q * CH + c] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(rr, 0u, 0u);
                        }
                }
                if (dmax > 0.5f - s3::kTieEps && F.dis_r64 != nullptr) {
                    unsigned tiemask = 0;
#pragma unroll
                    for (int q = 0; q < 4 * CH; ++q)
                        if (__builtin_fabsf(dist[q]) > 0.5f - s3::kTieEps) tiemask |= 1u << q;
                    while (tiemask != 0) {                                 // one round per queued output of the lane, not per output
                        const int q = __builtin_ctz(tiemask);
                        tiemask &= tiemask - 1;
                        const int c = q >> 2, r = (q >> 1) & 1, qq = q & 1;
                        const int il = il0 + r, xc = (jl0 + qq) * CH + c;
                        const int slot = atomicAdd(tq_count, 1);
                        if (slot < P.tq_cap) {
                            tq[slot] = ((uint32_t)il << 16) | (uint32_t)xc;
                            continue;
                        }
                        // queue full: evaluated on the spot (float64, the reference's own dtype chain), patched behind the store above
                        uint32_t dd[SS2];
                        double dx64[S], dy64[S];
#pragma unroll
                        for (int b = 0; b < S; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il) * S + b];
#pragma unroll
                        for (int a = 0; a < S; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl0 + qq) * S + a];
#pragma unroll
                        for (int a = 0; a < S; ++a)
#pragma unroll
                            for (int b = 0; b < S; ++b) dd[a * S + b] = dp[b * D::HP + a * CH + c];
                        outt[il * D::OUT_PITCH + ophase + xc] = (uint8_t)s3::resolve_u8<true, S>(dd, dx64, dy64, P.max_sigma);
                    }
                }
            }
            }
        };
        // ---- quad tasks (round 5): RP x CP neighbouring blocks of an INTERIOR x2 tile in one task.  Pair h + 1 starts exactly one
        //      tap right of / below pair h (ctl[28], ctl[29]), so the (S + RP - 1) x (S + CP - 1) taps of the task serve up to
        //      RP x CP blocks each: a tap's load and u8 -> f32 conversions once per task instead of once per block (S = 4,
        //      2 x 2 blocks: 25 instead of 64), its column-only terms once per column pair (40 instead of 64), its row
        //      products once per row pair.  Every output still sums ITS S x S taps in the order (a outer, b inner) with the
        //      operations of run_blocks: identical bytes.  One task per thread on a 64 x 64 tile with 2 x 2 blocks.
        //      Measured (A/B on one box, profiles/r05_experiments.txt): S = 4: 2 x 2 blocks +1.7 % (23.58 -> 23.99 Gpix/s; 2 x 1: +1.6 %,
        //      1 x 2: +1.2 %) -- the four v_exp_f32 per (tap, block) and their packed consumers are not shared, they are 3/4 of a
        //      tap's issue time; S = 2: 2 x 2 blocks -1.4 % (one 48-output task per thread hides less latency than four of 12), 2 x 1
        //      and 1 x 2 +-0: the 2 x 2 support keeps its single blocks.
#ifdef LERF_QUAD_RP
        constexpr int QRP = LERF_QUAD_RP, QCP = LERF_QUAD_CP;          // (A/B builds)
#else
        constexpr int QRP = S == 4 ? 2 : 1, QCP = S == 4 ? 2 : 1;
#endif
        const bool quad = BLK && QRP * QCP > 1 && blk_uni && __builtin_amdgcn_readfirstlane(ctl[28]) != 0 &&
                          __builtin_amdgcn_readfirstlane(ctl[29]) != 0 && ((nrow >> 1) % QRP) == 0 && ((ncol >> 1) % QCP) == 0;
        auto run_quads = [&]() {
            if constexpr (BLK && QRP * QCP > 1) {
            constexpr int RP = QRP, CP = QCP, TR = S + RP - 1, TC = S + CP - 1, SS2 = S * S;
            const int ncq = (ncol >> 1) / CP;
            const int ntq = ((nrow >> 1) / RP) * ncq;
            const unsigned magicq = (unsigned)((0x100000000ull + (unsigned)ncq - 1) / (unsigned)(ncq > 0 ? ncq : 1));
            for (int t = tid; t < ntq; t += NT) {
                const int g = ncq == 1 ? t : (int)__umulhi((unsigned)t, magicq);
                const int h = t - g * ncq;
                const int il0 = 2 * RP * g, jl0 = 2 * CP * h;
                const int lr = g_lr[il0], lc = g_lc[jl0];
                s3::f2 DX[RP][S], DY[CP][S];               // [pair][tap]: lanes = the two rows / the two columns of the pair
#pragma unroll
                for (int rp = 0; rp < RP; ++rp)
#pragma unroll
                    for (int b = 0; b < S; ++b) { DX[rp][b].x = g_dr[(il0 + 2 * rp) * S + b]; DX[rp][b].y = g_dr[(il0 + 2 * rp + 1) * S + b]; }
#pragma unroll
                for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                    for (int a = 0; a < S; ++a) { DY[cp][a].x = g_dc[(jl0 + 2 * cp) * S + a]; DY[cp][a].y = g_dc[(jl0 + 2 * cp + 1) * S + a]; }
                const uint32_t* dp = Dt + lr * D::HP + lc * CH;
                uint8_t* ob = outt + il0 * D::OUT_PITCH + ophase + jl0 * CH;
#pragma unroll 1
                for (int c = 0; c < CH; ++c) {
                    s3::f2 NUM[RP][CP][2], DEN[RP][CP][2];   // [row pair][column pair][row of the pair], lanes = columns of the pair
#pragma unroll
                    for (int A = 0; A < TC; ++A) {
#pragma unroll
                        for (int B = 0; B < TR; ++B) {
                            const uint32_t d = dp[B * D::HP + A * CH + c];
                            s3::f2 V, K1, K2, M2;
                            V.x = V.y = (float)(d >> 24);
                            K1.x = (float)((d >> 8) & 0xFFu);
                            M2.x = s3::gauss_m2rho_u8((float)(d & 0xFFu));
                            K2.x = (float)((d >> 16) & 0xFFu);
                            s3::f2 P0[CP], TY2[CP], TX[RP];
#pragma unroll
                            for (int cp = 0; cp < CP; ++cp) {
                                const int a = A - cp;
                                if (a >= 0 && a < S) {
                                    const s3::f2 TYV = s3::pk_mul_blo(DY[cp][a < 0 ? 0 : (a >= S ? 0 : a)], K2);
                                    P0[cp] = s3::pk_mul_blo(TYV, M2);
                                    TY2[cp] = s3::pk_mul(TYV, TYV);
                                }
                            }
#pragma unroll
                            for (int rp = 0; rp < RP; ++rp) {
                                const int b = B - rp;
                                if (b >= 0 && b < S) TX[rp] = s3::pk_mul_blo(DX[rp][b < 0 ? 0 : (b >= S ? 0 : b)], K1);
                            }
#pragma unroll
                            for (int cp = 0; cp < CP; ++cp) {
                                const int a = A - cp;
                                if (a < 0 || a >= S) continue;
#pragma unroll
                                for (int rp = 0; rp < RP; ++rp) {
                                    const int b = B - rp;
                                    if (b < 0 || b >= S) continue;
                                    const s3::f2 E0 = s3::pk_fma_ab<false>(TX[rp], P0[cp], s3::pk_fma_aa<false>(TX[rp], TY2[cp]));
                                    const s3::f2 E1 = s3::pk_fma_ab<true>(TX[rp], P0[cp], s3::pk_fma_aa<true>(TX[rp], TY2[cp]));
                                    s3::f2 W0, W1;
                                    W0.x = __builtin_amdgcn_exp2f(-E0.x); W0.y = __builtin_amdgcn_exp2f(-E0.y);
                                    W1.x = __builtin_amdgcn_exp2f(-E1.x); W1.y = __builtin_amdgcn_exp2f(-E1.y);
                                    // (compiler-generated consumers of the v_exp results: see run_blocks)
                                    if (a == 0 && b == 0) {
                                        DEN[rp][cp][0] = W0; DEN[rp][cp][1] = W1;
                                        NUM[rp][cp][0] = W0 * V; NUM[rp][cp][1] = W1 * V;
                                    } else {
                                        NUM[rp][cp][0] = __builtin_elementwise_fma(W0, V, NUM[rp][cp][0]);
                                        NUM[rp][cp][1] = __builtin_elementwise_fma(W1, V, NUM[rp][cp][1]);
                                        DEN[rp][cp][0] = DEN[rp][cp][0] + W0; DEN[rp][cp][1] = DEN[rp][cp][1] + W1;
                                    }
                                }
                            }
                        }
                    }
                    float dist[RP * CP * 4];
                    float dmax = 0.0f;
#pragma unroll
                    for (int rp = 0; rp < RP; ++rp)
#pragma unroll
                        for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                            for (int r = 0; r < 2; ++r)
#pragma unroll
                                for (int q = 0; q < 2; ++q) {
                                    const float den = q ? DEN[rp][cp][r].y : DEN[rp][cp][r].x, num = q ? NUM[rp][cp][r].y : NUM[rp][cp][r].x;
                                    const float xf = s3::finish_div(num, den);
                                    const float rr = __builtin_rintf(xf);
                                    const float ds = xf - rr;
                                    dist[((rp * CP + cp) * 2 + r) * 2 + q] = ds;
                                    dmax = __builtin_fmaxf(dmax, __builtin_fabsf(ds));
                                    ob[(2 * rp + r) * D::OUT_PITCH + (2 * cp + q) * CH + c] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(rr, 0u, 0u);
                                }
                    if (dmax > 0.5f - s3::kTieEps && F.dis_r64 != nullptr) {
                        unsigned tiemask = 0;
#pragma unroll
                        for (int u = 0; u < RP * CP * 4; ++u)
                            if (__builtin_fabsf(dist[u]) > 0.5f - s3::kTieEps) tiemask |= 1u << u;
                        while (tiemask != 0) {                             // one round per queued output of the lane, not per output
                            const int k = __builtin_ctz(tiemask);
                            tiemask &= tiemask - 1;
                            const int blkq = k >> 2, r = (k >> 1) & 1, q = k & 1;
                            const int rp = blkq / CP, cp = blkq - rp * CP;
                            const int il = il0 + 2 * rp + r, jl = jl0 + 2 * cp + q, xc = jl * CH + c;
                            const int slot = atomicAdd(tq_count, 1);
                            if (slot < P.tq_cap) {
                                tq[slot] = ((uint32_t)il << 16) | (uint32_t)xc;
                                continue;
                            }
                            // queue full: evaluated on the spot (float64, the reference's own dtype chain), patched behind the store above
                            uint32_t dd[SS2];
                            double dx64[S], dy64[S];
#pragma unroll
                            for (int b = 0; b < S; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il) * S + b];
#pragma unroll
                            for (int a = 0; a < S; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl) * S + a];
#pragma unroll
                            for (int a = 0; a < S; ++a)
#pragma unroll
                                for (int b = 0; b < S; ++b) dd[a * S + b] = Dt[(lr + rp + b) * D::HP + (lc + cp + a) * CH + c];
                            outt[il * D::OUT_PITCH + ophase + xc] = (uint8_t)s3::resolve_u8<true, S>(dd, dx64, dy64, P.max_sigma);
                        }
                    }
                }
            }
            }
        };
        // ---- LeRF-L block tasks (amplified-linear weights, 2 x 2 support): row groups and column groups of ONE or two
        //      members (x1.5 gives groups of 1, 2, 1, 2 ..., x2 pairs; the frame edges singles), one task = the up to 2 x 2
        //      outputs of a (row group, column group), all channels.  The dword-column tasks ran the general five-row form
        //      here (a group's size is not uniform at x1.5) and evaluated every tap's row factor per output; the block forms
        //      a tap's column factor once per column, its row factor once per row.  Weights as in lerf_stage3.h: the factor of
        //      a distance x is max(1 - alpha |x|, 0) for |x| <= 1 and 0 outside -- alpha * x + 1 (x < 0) and 1 - alpha * x
        //      (x >= 0) are the same float32 operations on |x| -- so the bytes are those of run_tasks.
        constexpr bool BLKL = KIND == LERF_KIND_LINEAR && S == 2 && D::OUT_FITS;
#ifndef LERF_NO_BLOCK_TASKS
        const bool blkl = BLKL && __builtin_amdgcn_readfirstlane((rows_align && ctl[26] != 0 && ctl[27] != 0 && (rowpitch & 15) == 0 &&
                                                                  nrow <= D::OUT_ROWS && ophase + ncolc <= D::OUT_PITCH) ? 1 : 0) != 0;
#else
        const bool blkl = false;
#endif
        auto run_blocks_lin = [&]() {
            if constexpr (BLKL) {
            const int* g_cgrp = reinterpret_cast<const int*>(smem + D::OFF_CGRP);
            const int ncg = __builtin_amdgcn_readfirstlane(ctl[25]);
            const int nblk = ngrp * ncg;
            const unsigned magicc = (unsigned)((0x100000000ull + (unsigned)ncg - 1) / (unsigned)(ncg > 0 ? ncg : 1));
            for (int t = tid; t < nblk; t += NT) {
                const int g = ncg == 1 ? t : (int)__umulhi((unsigned)t, magicc);
                const int h = t - g * ncg;
                const int il0 = g_grp[g], jl0 = g_cgrp[h];
                const bool two_r = g_grp[g + 1] - il0 > 1, two_c = g_cgrp[h + 1] - jl0 > 1;
                const int lr = g_lr[il0], lc = g_lc[jl0];
                // |distance| and inside-the-support mask (1 / 0) of the two rows [r][row tap b] and the two columns [q][column
                // tap a]; a single's second member reads its neighbour's entries (computed, never stored)
                float ax[2][2], mx[2][2], ay[2][2], my[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float x = g_dr[(il0 + r) * 2 + b];
                        ax[r][b] = __builtin_fabsf(x);
                        mx[r][b] = s3::dist_class_f(x) != 0 ? 1.0f : 0.0f;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const float y = g_dc[(jl0 + q) * 2 + a];
                        ay[q][a] = __builtin_fabsf(y);
                        my[q][a] = s3::dist_class_f(y) != 0 ? 1.0f : 0.0f;
                    }
                const uint32_t* dp = Dt + lr * D::HP + lc * CH;
                uint8_t* ob = outt + il0 * D::OUT_PITCH + ophase + jl0 * CH;
                float dist[4 * CH];                                  // [c][r][q]
                float dmax = 0.0f;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    float v[4], fy[2][4], fx[2][4];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const uint32_t d = dp[b * D::HP + a * CH + c];
                            v[a * 2 + b] = (float)(d >> 24);
                            const float alpha = s3::lin_alpha_u8((float)(d & 0xFFu), ms255);
#pragma unroll
                            for (int q = 0; q < 2; ++q) fy[q][a * 2 + b] = s3::lin_factor_abs(alpha, ay[q][a], my[q][a]);
#pragma unroll
                            for (int r = 0; r < 2; ++r) fx[r][a * 2 + b] = s3::lin_factor_abs(alpha, ax[r][b], mx[r][b]);
                        }
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float e[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) e[k] = fx[r][k] * fy[q][k];
                            const float xf = s3::finish<false, 4, true, true, true>(e, v);
                            const float rr = __builtin_rintf(xf);
                            const bool have = (r == 0 || two_r) && (q == 0 || two_c);
                            const float ds = have ? xf - rr : 0.0f;
                            dist[(c * 2 + r) * 2 + q] = ds;
                            dmax = __builtin_fmaxf(dmax, __builtin_fabsf(ds));
                            if (have) ob[r * D::OUT_PITCH + q * CH + c] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(rr, 0u, 0u);
                        }
                }
                if (dmax > 0.5f - s3::kTieEps && F.dis_r64 != nullptr) {
                    unsigned tiemask = 0;
#pragma unroll
                    for (int q = 0; q < 4 * CH; ++q)
                        if (__builtin_fabsf(dist[q]) > 0.5f - s3::kTieEps) tiemask |= 1u << q;
                    while (tiemask != 0) {                                 // one round per queued output of the lane, not per output
                        const int q = __builtin_ctz(tiemask);
                        tiemask &= tiemask - 1;
                        const int c = q >> 2, r = (q >> 1) & 1, qq = q & 1;
                        const int il = il0 + r, xc = (jl0 + qq) * CH + c;
                        const int slot = atomicAdd(tq_count, 1);
                        if (slot < P.tq_cap) {
                            tq[slot] = ((uint32_t)il << 16) | (uint32_t)xc;
                            continue;
                        }
                        uint32_t dd[4];
                        double dx64[2], dy64[2];
#pragma unroll
                        for (int b = 0; b < 2; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il) * 2 + b];
#pragma unroll
                        for (int a = 0; a < 2; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl0 + qq) * 2 + a];
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) dd[a * 2 + b] = dp[b * D::HP + a * CH + c];
                        outt[il * D::OUT_PITCH + ophase + xc] = (uint8_t)s3::resolve_u8<false, 2>(dd, dx64, dy64, P.max_sigma);
                    }
                }
            }
            }
        };
        // the LDS image of the output block -> the frame: whole 16-byte chunks with streaming stores, the ragged ends of a
        // row (the neighbouring tiles' bytes share those chunks) dword by dword and byte by byte
        auto flush_blocks = [&]() {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const int nchunk = (ophase + ncolc + 15) >> 4;
            const unsigned magicf = (unsigned)((0x100000000ull + (unsigned)nchunk - 1) / (unsigned)nchunk);
            uint8_t* gbase = seg0 - ophase;
            const int total = nrow * nchunk;
            const int full_lo = ophase == 0 ? 0 : 1, full_hi = ((ophase + ncolc) & 15) == 0 ? nchunk : nchunk - 1;   // whole chunks of a row
            for (int i = tid; i < total; i += NT) {
                const int row = nchunk == 1 ? i : (int)__umulhi((unsigned)i, magicf);
                const int ch = i - row * nchunk;
                if (ch >= full_lo && ch < full_hi) {
                    const v4u x = *reinterpret_cast<const v4u*>(outt + row * D::OUT_PITCH + ch * 16);
                    uint8_t* dst = gbase + row * rowpitch + ch * 16;
                    // streaming stores for the 64-byte lines this tile writes completely; the first and the last line of a row
                    // are shared with the neighbouring tiles and go through the L2 (regular stores), which merges the two
                    // tiles' halves into one line -- as streaming stores they were two partial-line writes each (+67 MB per step)
                    const uintptr_t line = reinterpret_cast<uintptr_t>(dst) & ~(uintptr_t)63;
                    const uintptr_t r0 = reinterpret_cast<uintptr_t>(gbase + row * rowpitch) + (uintptr_t)ophase;
                    if (line >= r0 && line + 64 <= r0 + (uintptr_t)ncolc) __builtin_nontemporal_store(x, reinterpret_cast<v4u*>(dst));
                    else *reinterpret_cast<v4u*>(dst) = x;
                }
            }
            // the ragged chunks, two per row at most, in a loop of their own (a few waves take it once, instead of every
            // wave dragging the byte path through every round of the loop above)
            for (int i = tid; i < 2 * nrow; i += NT) {
                const int row = i >> 1;
                const int ch = (i & 1) ? nchunk - 1 : 0;
                if (ch >= full_lo && ch < full_hi) continue;             // that end of the row is a whole chunk
                if ((i & 1) && nchunk == 1) continue;                    // a one-chunk row: its first-chunk lane has it
                const uint8_t* src = outt + row * D::OUT_PITCH + ch * 16;
                uint8_t* dst = gbase + row * rowpitch + ch * 16;
                const int lo = max(ophase - ch * 16, 0), hi = min(ophase + ncolc - ch * 16, 16);
                const v4u x = *reinterpret_cast<const v4u*>(src);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w = x[k];
                    if (lo <= 4 * k && hi >= 4 * k + 4) {
                        *reinterpret_cast<uint32_t*>(dst + 4 * k) = w;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (4 * k + u >= lo && 4 * k + u < hi) dst[4 * k + u] = (uint8_t)(w >> (8 * u));
                    }
                }
            }
        };
        // the constant-size variants cover the integer scale factors (x2 -> 2 rows per group, ...); anything else reads
        // the group size per task
        constexpr bool WIDE = KIND == LERF_KIND_GAUSS && S == 2;      // x3 / x4 variants where the registers allow
        // (max_sigma <= s3::kNoShiftMaxSigma here: the host sends larger values to the float64 direct kernel, lerf_fused.hip)
        if (quad) run_quads();
        else if (blk_uni) run_blocks(std::true_type{});
        else if (blk) run_blocks(std::false_type{});
        else if (blkl) run_blocks_lin();
        else if (gsame == 2) run_tasks(std::integral_constant<int, 2>{});
        else if (WIDE && gsame == 3) run_tasks(std::integral_constant<int, WIDE ? 3 : 0>{});
        else if (WIDE && gsame == 4) run_tasks(std::integral_constant<int, WIDE ? 4 : 0>{});
        else run_tasks(std::integral_constant<int, 0>{});
        // ---- tie pass: the queued outputs in float64, one per lane; each patches its byte behind the task loop's
        //      dword stores (drained and fenced by the barrier)
        if (blk || blkl) {
            // block tasks: the queued outputs are patched in the LDS image, which then leaves in one piece
            __syncthreads();
            LERF_STAMP(15);
            if (F.dis_r64 != nullptr) {
                const int nq = min(*tq_count, P.tq_cap);
                if (nq > 0) {
                    for (int i = tid; i < nq; i += NT) {
                        const uint32_t e = tq[i];
                        const int il = (int)(e >> 16), xc = (int)(e & 0xFFFFu);
                        const int jl = xc / CH;
                        const int c = xc - jl * CH;
                        const int lr = g_lr[il], lc = g_lc[jl];
                        uint32_t dd[SS];
                        double dx64[S], dy64[S];
#pragma unroll
                        for (int b = 0; b < S; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il) * S + b];
#pragma unroll
                        for (int a = 0; a < S; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl) * S + a];
#pragma unroll
                        for (int a = 0; a < S; ++a)
#pragma unroll
                            for (int b = 0; b < S; ++b) dd[a * S + b] = Dt[(lr + b) * D::HP + (lc + a) * CH + c];
                        outt[il * D::OUT_PITCH + ophase + xc] = (uint8_t)s3::resolve_u8<KIND == LERF_KIND_GAUSS, S>(dd, dx64, dy64, P.max_sigma);
                    }
                    __syncthreads();
                }
            }
            flush_blocks();
        } else if (F.dis_r64 != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int nq = min(*tq_count, P.tq_cap);
            for (int i = tid; i < nq; i += NT) {
                const uint32_t e = tq[i];
                const int il = (int)(e >> 16), xc = (int)(e & 0xFFFFu);
                const int jl = xc / CH;
                const int c = xc - jl * CH;
                const int lr = g_lr[il], lc = g_lc[jl];
                uint32_t dd[SS];
                double dx64[S], dy64[S];
#pragma unroll
                for (int b = 0; b < S; ++b) dx64[b] = F.dis_r64[(int64_t)(i0 + il) * S + b];
#pragma unroll
                for (int a = 0; a < S; ++a) dy64[a] = F.dis_c64[(int64_t)(j0 + jl) * S + a];
#pragma unroll
                for (int a = 0; a < S; ++a)
#pragma unroll
                    for (int b = 0; b < S; ++b) dd[a * S + b] = Dt[(lr + b) * D::HP + (lc + a) * CH + c];
                seg0[il * rowpitch + xc] = (uint8_t)s3::resolve_u8<KIND == LERF_KIND_GAUSS, S>(dd, dx64, dy64, P.max_sigma);
            }
        }
    }
#ifdef LERF_STAMPS
    __syncthreads();
    LERF_STAMP(12);
    LERF_STAMP_RT(14);
#endif
}

// ---------------------------------------------------------------------------
// stage 1 alone, one 64x64 block per workgroup with no halo (two-launch path): input tile (block + 3 px) -> the
// three byte LUTs, 4 rotations each -> feat bytes to P.feat.  Same phases as in sr_fused_kernel.
// ---------------------------------------------------------------------------
constexpr int up16c(int x) { return (x + 15) / 16 * 16; }
struct DimsA {
    static constexpr int FY = TH, FX = TW, FP = FX * CH, NF = FY * FP;
    static constexpr int IY = FY + 2 * R1, IX = FX + 2 * R1, IPB = IX * CH, IP = (IPB + 3 + 3) / 4 * 4, NI = IY * IP;
    static constexpr int OFF_LUT = 0;
    static constexpr int OFF_C = LUT_PAD;
    static constexpr int OFF_ACC = OFF_C + up16c(NI);
    static constexpr int OFF_F = OFF_ACC + up16c(NF * 2);
    static constexpr int LDS_BYTES = OFF_F + up16c(NF);
};

template <bool GEN = false>
__global__ void __launch_bounds__(NT)
s1_kernel(Params P, std::conditional_t<GEN, RaggedTable, NoTable> T) {
    using D = DimsA;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    int bid = xcd_order((int)blockIdx.x, (int)gridDim.x);
    const FrameView F = frame_view<GEN>(P, T, bid);
    const int tyi = bid / F.tiles_x, txi = bid - tyi * F.tiles_x;
    // (P.ty_org, P.tx_org) = origin of the stage-1 region: (0, 0) for whole frames, the region of interest widened by the
    // reach of stages 2 + 3 for a rank's block (launch_fused_t)
    const int fy0 = P.ty_org + tyi * TH, fx0 = P.tx_org + txi * TW;
    const int iy0 = fy0 - R1, ix0 = fx0 - R1;
    const int H = F.H, W = F.W;
    const uint8_t* __restrict__ img = F.img;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + D::IY <= H && ix0 + D::IX <= W;
    const int Hc = interior ? -1 : H, Wc = W;
    uint8_t* Ft = smem + D::OFF_F;
    LERF_STAMP(0);

#ifndef LERF_S1_LDS_PIXELS
    // interior tiles of the specialised kernel: pixel reads on the vector-memory path (byte_phase_vmem), no input tile in LDS
    const bool vmem_pixels = !GEN && interior && !P.host_input;      // pinned host frames (stream.StreamingSR) cross PCIe uncached
#else
    const bool vmem_pixels = false;
#endif
    if (vmem_pixels) {
        int8_t* lut = reinterpret_cast<int8_t*>(smem + D::OFF_LUT);
        int16_t* acc = reinterpret_cast<int16_t*>(smem + D::OFF_ACC);
        const int div1 = kQ * 3;
        const uint8_t* org = img + ((int64_t)fy0 * W + fx0) * CH;
        const int pitch = W * CH;
        constexpr int L1N = (LERF_LUT_ENTRIES + 15) / 16, L1TAIL = L1N - 5 * NT;
        uint4 n0, n1, n2, n3, n4, n5 = make_uint4(0, 0, 0, 0);
        copy16<LERF_LUT_ENTRIES>(smem + D::OFF_LUT, P.pack + 0 * LUT_PAD, tid);
        __syncthreads();
        LERF_S1_LOAD(P.pack + 1 * LUT_PAD);
        byte_phase_vmem<D::NF, D::FP, 's', 0, 4, 1, 0>(lut, org, pitch, acc, Ft, div1, 0, tid);
        __syncthreads();
        LERF_S1_STORE();
        __syncthreads();
        LERF_S1_LOAD(P.pack + 2 * LUT_PAD);
        byte_phase_vmem<D::NF, D::FP, 'c', 0, 4, 1, 1>(lut, org, pitch, acc, Ft, div1, 0, tid);
        __syncthreads();
        LERF_S1_STORE();
        __syncthreads();
        byte_phase_vmem<D::NF, D::FP, 't', 0, 4, 1, 2>(lut, org, pitch, acc, Ft, div1, 0, tid);
        __syncthreads();
    } else {
    const int cphase = load_input_tile<D::IY, D::IPB, D::IP>(smem + D::OFF_C, img, H, W, iy0, ix0, interior, tid, []() {});
    {
        const uint8_t* Ct = smem + D::OFF_C + cphase;
        int8_t* lut = reinterpret_cast<int8_t*>(smem + D::OFF_LUT);
        int16_t* acc = reinterpret_cast<int16_t*>(smem + D::OFF_ACC);
        const int div1 = kQ * 3;
        constexpr int L1N = (LERF_LUT_ENTRIES + 15) / 16, L1TAIL = L1N - 5 * NT;
        uint4 n0, n1, n2, n3, n4, n5 = make_uint4(0, 0, 0, 0);
        copy16<LERF_LUT_ENTRIES>(smem + D::OFF_LUT, P.pack + 0 * LUT_PAD, tid);
        __syncthreads();
        if constexpr (GEN) {
            const int dv = kQ * P.n1;
            // interior tiles: neighbourhood pixels over the vector-memory path (byte_phase_vmem_rt), like the specialised kernel
            const bool vmg = interior && !P.host_input;
            const uint8_t* vorg = img + ((int64_t)fy0 * W + fx0) * CH;
#pragma unroll 1
            for (int m = 0; m < P.n1; ++m) {
                const bool more = m + 1 < P.n1;
                if (more) LERF_S1_LOAD(P.pack + (size_t)(m + 1) * LUT_PAD);
                if (vmg) byte_phase_vmem_rt<D::NF, D::FP, 4>(lut, vorg, W * CH, acc, Ft, dv, 0, tid, P.s1dyx[m], m == 0, !more);
                else
                byte_phase_rt<D::NF, D::FP, D::IP, 4, true>(lut, Ct, acc, Ft, fy0, fx0, iy0, ix0, Hc, Wc, dv, 0, tid, P.s1off[m], m == 0, !more);
                __syncthreads();
                if (more) {
                    LERF_S1_STORE();
                    __syncthreads();
                }
            }
        } else {
        LERF_S1_LOAD(P.pack + 1 * LUT_PAD);
        byte_phase<D::NF, D::FP, D::IP, 's', 0, 4, 1, 0, true>(lut, Ct, acc, Ft, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        LERF_S1_STORE();
        __syncthreads();
        LERF_S1_LOAD(P.pack + 2 * LUT_PAD);
        byte_phase<D::NF, D::FP, D::IP, 'c', 0, 4, 1, 1, true>(lut, Ct, acc, Ft, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        LERF_S1_STORE();
        __syncthreads();
        byte_phase<D::NF, D::FP, D::IP, 't', 0, 4, 1, 2, true>(lut, Ct, acc, Ft, fy0, fx0, iy0, ix0, Hc, Wc, div1, 0, tid);
        __syncthreads();
        }
    }
    }
    LERF_STAMP(1);
    // the block's rows to P.feat
    uint8_t* __restrict__ fdst = F.feat;
    const int rows = min(TH, H - fy0), cols3 = min(TW, W - fx0) * CH;
    const bool dwords = cols3 == D::FP && ((W * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(F.feat) & 3) == 0;
    if (dwords) {
        constexpr int RD = D::FP / 4;
        uint32_t* dst = reinterpret_cast<uint32_t*>(fdst + ((int64_t)fy0 * W + fx0) * CH);
        const int rowdw = (W * CH) >> 2;
        for (int i = tid; i < rows * RD; i += NT) {
            const int r = i / RD;
            __builtin_nontemporal_store(reinterpret_cast<const uint32_t*>(Ft)[i], &dst[(int64_t)r * rowdw + (i - r * RD)]);
        }
    } else {
        for (int i = tid; i < rows * cols3; i += NT) {
            const int r = i / cols3, c3 = i - r * cols3;
            fdst[((int64_t)(fy0 + r) * W + fx0) * CH + c3] = Ft[r * D::FP + c3];
        }
    }
#ifdef LERF_STAMPS
    __syncthreads();
    LERF_STAMP(2);
#endif
}

#undef LERF_S1_LOAD
#undef LERF_S1_STORE
#undef LERF_ADDTID
#undef LERF_ADDTID4
#undef LERF_SET_M0


// ---------------------------------------------------------------------------
// host side of this instance
// ---------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device instead of once per launch
template <auto KERN>
static int ensure_lds(int bytes) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LERF_ELAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return LERF_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return LERF_ELAUNCH;
    done.fetch_or(bit, std::memory_order_relaxed);
    return LERF_OK;
}

inline size_t feat_slice_bytes(int H, int W) { return ((size_t)H * W * CH + 15) / 16 * 16; }

template <int S, int KIND, bool EMIT, bool GEN>
static int launch_fused_t(const FusedArgs& a, hipStream_t st) {
    using D = Dims<S, GEN, EMIT>;
    using TableT = std::conditional_t<GEN, RaggedTable, NoTable>;
    const lerf_luts_t* L = a.luts;
    Params P{};
    P.tq_cap = a.tq_cap == 0 ? D::TQ_CAP : (a.tq_cap < 0 ? 0 : (a.tq_cap > D::TQ_CAP ? D::TQ_CAP : a.tq_cap));
    P.pad_mode = a.pad_mode;
    P.host_input = a.host_input ? 1 : 0;
    P.out_pitch = a.out_pitch;
    P.img = a.img; P.in_sn = a.in_sn; P.out = a.out; P.out_sn = a.out_sn;
    P.H = a.H; P.W = a.W; P.oH = a.oH; P.oW = a.oW;
    const bool roi = a.roi_h > 0 && a.roi_w > 0;
    P.ty_org = roi ? a.roi_y : 0; P.tx_org = roi ? a.roi_x : 0;
    P.tiles_y = ((roi ? a.roi_h : a.H) + TH - 1) / TH;
    P.tiles_x = ((roi ? a.roi_w : a.W) + TW - 1) / TW;
    P.pack = (const uint8_t*)L->fused_pack;
    P.left_r = a.left_r; P.dis_r = a.dis_r; P.left_c = a.left_c; P.dis_c = a.dis_c;
    P.dis_r64 = a.dis_r64; P.dis_c64 = a.dis_c64;
    P.max_sigma = a.max_sigma;
    P.n1 = L->n_modes1; P.n2 = L->n_modes2;
    P.stamps = EMIT ? nullptr : (unsigned long long*)a.workspace;
    for (int l = 0; l < 2 * P.n2; ++l) {
        const char mc = L->modes2[l >> 1];
        const Off3 o0 = tile_offsets<D::FP>(mc, l & 1), o1 = tile_offsets<D::FP>(mc, (l & 1) + 2);
        for (int i = 0; i < 3; ++i) {
            P.s2off[l][i] = o0.o[i];
            P.s2off[l][3 + i] = o1.o[i];
        }
    }
    for (int m = 0; m < P.n1; ++m)
        for (int r = 0; r < 4; ++r) {
            const DyDx q = pattern_dydx(L->modes1[m], r);
            for (int i = 0; i < 3; ++i) P.s1dyx[m][3 * r + i] = (q.dy[i] << 16) | (q.dx[i] & 0xFFFF);
        }
    for (int l = 0; l < 2 * P.n2; ++l)
        for (int h = 0; h < 2; ++h) {
            const DyDx q = pattern_dydx(L->modes2[l >> 1], (l & 1) + 2 * h);
            for (int i = 0; i < 3; ++i) P.s2dyx[l][3 * h + i] = (q.dy[i] << 16) | (q.dx[i] & 0xFFFF);
        }
    P.emit = (uint32_t*)a.emit; P.emit_sn = a.emit_sn;
    P.feat = nullptr; P.feat_sn = 0;
    if constexpr ((KIND & LERF_FUSED_WARP) != 0) {
        if (!a.wgeo || !a.wboxes || a.workspace == nullptr || a.roi_h > 0 || a.items != nullptr) return LERF_EINVAL;
        P.wgeo = *a.wgeo;
        P.wboxes = a.wboxes;
    }
    auto set_s1off = [&](Params& Q, int pitch_kind) {          // stage-1 pattern offsets in the input tile of the kernel that runs it
        for (int m = 0; m < Q.n1; ++m)
            for (int r = 0; r < 4; ++r) {
                const Off3 o = pitch_kind == 0 ? tile_offsets<D::IP>(L->modes1[m], r) : tile_offsets<DimsA::IP>(L->modes1[m], r);
                for (int i = 0; i < 3; ++i) Q.s1off[m][3 * r + i] = o.o[i];
            }
    };
    set_s1off(P, 0);
    bool two = a.workspace != nullptr && !(a.flags & LERF_GEO_SINGLE_LAUNCH);
#ifdef LERF_STAMPS
    // diagnostic build: the stamps live at the START of the workspace -- [n x tiles x 16] of the stages-2+3 (or single) launch,
    // then [n x stage-1 blocks x 16] of s1_kernel, then (two launches) the stage-1 output.  A workspace too small for
    // all three keeps the single launch (tools/stamps.py sizes it).
    size_t stamp_bytes = 0;
    {
        const size_t per = 16 * sizeof(unsigned long long);
        const size_t nb = (size_t)a.n * P.tiles_y * P.tiles_x;
        stamp_bytes = ((2 * nb + nb / 4 + 64) * per + 255) & ~(size_t)255;   // both areas (the stage-1 grid of a region is a little larger)
        two = two && a.items == nullptr && !EMIT && a.workspace_bytes >= stamp_bytes + (size_t)a.n * feat_slice_bytes(a.H, a.W);
    }
#endif
    auto ka = s1_kernel<GEN>;
    auto kb = sr_fused_kernel<S, KIND, EMIT, true, GEN>;
    auto kc = sr_fused_kernel<S, KIND, EMIT, false, GEN>;
    int rc;
    if (two) {
        if ((rc = ensure_lds<s1_kernel<GEN>>(DimsA::LDS_BYTES)) != LERF_OK) return rc;
        if ((rc = ensure_lds<sr_fused_kernel<S, KIND, EMIT, true, GEN>>(D::LDS_BYTES)) != LERF_OK) return rc;
    } else if ((rc = ensure_lds<sr_fused_kernel<S, KIND, EMIT, false, GEN>>(D::LDS_BYTES)) != LERF_OK) return rc;

    if constexpr (GEN) {
        if (a.items != nullptr) {
            // frames of different sizes: chunks of RAGGED_MAX descriptors per launch (pair)
            size_t ws_off = 0;
            for (int i0 = 0; i0 < a.n_items; i0 += RAGGED_MAX) {
                RaggedTable T{};
                T.n = a.n_items - i0 < RAGGED_MAX ? a.n_items - i0 : RAGGED_MAX;
                int64_t blocks = 0;
                for (int k = 0; k < T.n; ++k) {
                    const FusedItem& it = a.items[i0 + k];
                    FrameDesc& d = T.d[k];
                    d.img = it.img; d.out = it.out; d.emit = (uint32_t*)it.emit;
                    d.left_r = it.left_r; d.dis_r = it.dis_r; d.left_c = it.left_c; d.dis_c = it.dis_c;
                    d.dis_r64 = (it.dis_r64 && it.dis_c64) ? it.dis_r64 : nullptr;
                    d.dis_c64 = (it.dis_r64 && it.dis_c64) ? it.dis_c64 : nullptr;
                    d.H = it.H; d.W = it.W; d.oH = it.oH; d.oW = it.oW;
                    d.tiles_x = (it.W + TW - 1) / TW; d.tiles_y = (it.H + TH - 1) / TH;
                    d.first_block = (int)blocks;
                    blocks += (int64_t)d.tiles_x * d.tiles_y;
                    d.feat = two ? (uint8_t*)a.workspace + ws_off : nullptr;
                    if (two) ws_off += feat_slice_bytes(it.H, it.W);
                }
                if (blocks > 0x7FFFFFFF) return LERF_EUNSUPPORTED;
                // every tile of the tie guard needs both float64 tables: the kernel tests P.dis_r64 (set per frame)
                if (two) {
                    Params Pa = P;
                    set_s1off(Pa, 1);
                    hipLaunchKernelGGL(ka, dim3((unsigned)blocks), dim3(NT), DimsA::LDS_BYTES, st, Pa, T);
                    hipLaunchKernelGGL(kb, dim3((unsigned)blocks), dim3(NT), D::LDS_BYTES, st, P, T);
                } else {
                    hipLaunchKernelGGL(kc, dim3((unsigned)blocks), dim3(NT), D::LDS_BYTES, st, P, T);
                }
            }
            return LERF_OK;
        }
    }
    const int64_t blocks = (int64_t)a.n * P.tiles_y * P.tiles_x;
    if (blocks > 0x7FFFFFFF) return LERF_EUNSUPPORTED;
    TableT T{};
    // two launches when the caller's workspace can hold the stage-1 output of the batch: stage 1 without halo
    // recomputation, then stages 2+3 from it
    if (two) {
        P.feat = (uint8_t*)a.workspace;
#ifdef LERF_STAMPS
        P.feat += stamp_bytes;
#endif
        P.feat_sn = (int64_t)feat_slice_bytes(a.H, a.W);
        Params Pa = P;
#ifdef LERF_STAMPS
        Pa.stamps = P.stamps + (size_t)blocks * 16;
#endif
        set_s1off(Pa, 1);
        int64_t blocks1 = blocks;
        if (roi) {
            // region of interest: stage 1 over the region widened by the reach of stages 2 + 3 (whole 64 x 64 blocks from an
            // origin on a 4-pixel boundary, so that the feat rows keep their aligned dword stores; a block that reaches past
            // the widened region computes valid stage-1 values nobody reads)
            const int reach = R2 + D::R3;
            const int y0 = a.roi_y - reach > 0 ? a.roi_y - reach : 0, x0 = a.roi_x - reach > 0 ? (a.roi_x - reach) & ~3 : 0;
            const int y1 = a.roi_y + a.roi_h + reach < a.H ? a.roi_y + a.roi_h + reach : a.H;
            const int x1 = a.roi_x + a.roi_w + reach < a.W ? a.roi_x + a.roi_w + reach : a.W;
            Pa.ty_org = y0; Pa.tx_org = x0;
            Pa.tiles_y = (y1 - y0 + TH - 1) / TH; Pa.tiles_x = (x1 - x0 + TW - 1) / TW;
            blocks1 = (int64_t)a.n * Pa.tiles_y * Pa.tiles_x;
            if (blocks1 > 0x7FFFFFFF) return LERF_EUNSUPPORTED;
        }
        hipLaunchKernelGGL(ka, dim3((unsigned)blocks1), dim3(NT), DimsA::LDS_BYTES, st, Pa, T);
        hipLaunchKernelGGL(kb, dim3((unsigned)blocks), dim3(NT), D::LDS_BYTES, st, P, T);
        return LERF_OK;
    }
    hipLaunchKernelGGL(kc, dim3((unsigned)blocks), dim3(NT), D::LDS_BYTES, st, P, T);
    return LERF_OK;
}

// dispatch over (support, kind) for one (EMIT, GEN) family of this instance
template <bool GEN>
static int launch_sr(const FusedArgs& a, hipStream_t st) {
    if (a.kind == LERF_KIND_GAUSS) {
        if (a.S == 2) return launch_fused_t<2, LERF_KIND_GAUSS, false, GEN>(a, st);
        if (a.S == 4) return launch_fused_t<4, LERF_KIND_GAUSS, false, GEN>(a, st);
    } else if (a.kind == LERF_KIND_LINEAR) {
        if (a.S == 2) return launch_fused_t<2, LERF_KIND_LINEAR, false, GEN>(a, st);
    }
    return LERF_EUNSUPPORTED;
}
#ifdef LERF_FUSED_WITH_WARP
static int launch_warp(const FusedArgs& a, hipStream_t st) {
    if (a.kind == LERF_KIND_GAUSS) return launch_fused_t<2, LERF_KIND_GAUSS | LERF_FUSED_WARP, false, false>(a, st);
    if (a.kind == LERF_KIND_LINEAR) return launch_fused_t<2, LERF_KIND_LINEAR | LERF_FUSED_WARP, false, false>(a, st);
    return LERF_EUNSUPPORTED;
}
#endif
template <bool GEN>
static int launch_stages(const FusedArgs& a, hipStream_t st) {
    if (a.luts->oC == 3) return launch_fused_t<2, LERF_KIND_GAUSS, true, GEN>(a, st);
    if (a.luts->oC == 1) return launch_fused_t<2, LERF_KIND_LINEAR, true, GEN>(a, st);
    return LERF_EUNSUPPORTED;
}

}  // namespace LERF_FUSED_NS
}  // namespace lerf
