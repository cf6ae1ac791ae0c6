/*
 * hbs_common.h -- shared definitions of the HIP kernels and their host shim.
 *
 * The per-tile logic in hbs_tile.h / hbs_emit.h is written as plain C++ that
 * hipcc compiles for gfx950 and that tests/sim/ can also compile with g++ to
 * single-step the same code on the CPU against the oracle (HBS_HOST_SIM).
 * That CPU build exists only under tests/; the product library contains the
 * device code alone and never falls back to it.
 */
#ifndef HBS_COMMON_H
#define HBS_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/hevcbitstream_amd.h"

#if defined(__HIPCC__) && !defined(HBS_HOST_SIM)
#define HBS_HD __host__ __device__ __forceinline__
#define HBS_D  __device__ __forceinline__
#define HBS_M  __host__ __device__ __forceinline__      /* member functions */
#else
#define HBS_HD static inline
#define HBS_D  static inline
#define HBS_M  inline
#endif

namespace hbs {

/* ---- geometry of the scan/extract kernel --------------------------------- */
constexpr int kThreads         = 512;            /* 8 wavefronts per workgroup          */
constexpr int kWaves           = kThreads / 64;
constexpr int kBlockBytes      = 64;             /* unit of classification (one u64 mask) */
constexpr int kBlocksPerThread = 2;              /* contiguous blocks per thread        */
constexpr int kThreadBytes     = kBlockBytes * kBlocksPerThread;     /* 128 B           */
constexpr int kTileBytes       = kThreads * kThreadBytes;            /* 64 KiB per tile */
constexpr int kBlocks          = kThreads * kBlocksPerThread;        /* 1024 per tile   */
constexpr int kPad             = 256;            /* image bytes kept before/after the tile */
constexpr int kImageBytes      = kPad + kTileBytes + kPad;

struct alignas(16) Quad { uint32_t x, y, z, w; };

/*
 * View of the tile image in LDS.  Logical offset o (bytes, relative to the
 * tile's first stream byte, o in [-kPad, kTileBytes + kPad)) lives at phys(o).
 * The image is XOR-swizzled at 16-byte granularity inside each 256-byte row
 * (slot ^= row & 15) so that both access shapes are bank-conflict free:
 *   - staging: consecutive lanes write consecutive 16-byte slots;
 *   - classification: lane t reads its own 128 contiguous bytes (8 slots).
 */
struct TileView {
    const uint8_t* img;
    HBS_M static uint32_t phys(int32_t o)
    {
        const uint32_t a = (uint32_t)(o + kPad);
        uint32_t slot = a >> 4;
        slot ^= (slot >> 4) & 15u;
        return (slot << 4) | (a & 15u);
    }
    HBS_M uint32_t byte(int32_t o) const { return img[phys(o)]; }
    HBS_M uint32_t dword(int32_t o) const { return *reinterpret_cast<const uint32_t*>(img + phys(o)); }   /* o % 4 == 0 */
    HBS_M Quad quad(int32_t o) const { return *reinterpret_cast<const Quad*>(img + phys(o)); }            /* o % 16 == 0 */
};

/* state carried between tiles: are we inside a NAL payload? */
enum : uint32_t { kKindNone = 0, kKindStart = 1, kKindStop = 2 };

/* Device-side header: results of the main kernel handed to the finalize
 * kernels.  Zeroed (memset) before every launch except first_empty. */
struct RunHeader {
    uint64_t final_kept;      /* RBSP bytes produced by the tile passes          */
    uint64_t final_nals;      /* valid start codes seen by the tile passes       */
    uint32_t final_inside;    /* state after the last stream byte                */
    uint32_t error;           /* HBS_E_* (positive magnitude) or 0               */
    unsigned long long first_empty;  /* ordinal of the first empty NAL (min), init ~0 */
    uint32_t abort_flag;      /* set when a look-back wait timed out             */
    uint32_t probe_chunks;    /* density probe: 16-byte chunks sampled ...                      */
    uint32_t probe_flagged;   /* ... and how many of them may hold a 00 00 pair (hbs_sparse.h)  */
    /* dense tiles counted ahead of the event-sparse kernel (hbs_scan4.hip, round 5): this call's stamp, the word per tile the
     * prologue's sample and k_scan_ahead4 leave (stamp | 1 marked, stamp | 2 counted) and the table of the tiles counted; set by
     * the prologue of every call (ahead_tab = 0: not this call) */
    uint32_t pad_a;
    unsigned long long ahead_cand, ahead_tab, ahead_stamp;
    uint32_t rewalk_count;    /* index-only scan (hbs_scan5.hip): tiles walked by rows, listed for the emit pass's helpers; cleared by the prologue */
    uint32_t pad0[13];
    uint32_t ticket;          /* next unclaimed tile (dynamic tile schedules); alone on its 128-byte line */
    uint32_t pad1[31];
    uint32_t probe_slot[64][2];   /* density probe, one pair per probe workgroup: chunks sampled, chunks flagged.  Plain
                                     stores by the prologue kernel (no zeroing needed in front); readers add them up */
};
static_assert(sizeof(RunHeader) == 768, "RunHeader layout");

/* What the count-ahead keeps between calls, in device memory and written by kernels only, so that a call captured into a HIP
 * graph and replayed behaves like a call made again: `call` numbers the calls that used the workspace (the last launch of a
 * call, k_scan_finish, adds one: the next call's stamp is another one, and the words the tiles carry from this call mean nothing
 * to it), `listed` counts the tiles the prologue's sample marked in the call in progress (k_scan_finish clears it). */
struct AheadCtl { unsigned long long call; uint32_t listed; uint32_t pad[13]; };
static_assert(sizeof(AheadCtl) == 64, "one line");
HBS_HD unsigned long long ahead_stamp_of(unsigned long long call) { return (call + 1ull) << 2; }

/* The kernel-choice rule of the automatic mode (hbs_scan.hip launch_scan_extract): the event-sparse
 * kernel handles flagged chunks 64 at a time on one wavefront, so once more than one chunk in
 * kDenseOneIn may hold a zero pair the LDS-image kernel, whose cost does not depend on the data,
 * is the faster of the two. */
#ifndef HBS_DENSE_ONE_IN
#define HBS_DENSE_ONE_IN 26
#endif
constexpr uint32_t kDenseOneIn = HBS_DENSE_ONE_IN;    /* Round 3: the probe counts ELEMENTS (chunks a pattern 00 00 {<=3} ends in, neighbours seen), and the event-sparse
                                           kernel keeps 0.42-0.53 of peak up to the density at which tiles pass kDenseElems = 512 of 12288 chunks
                                           (4.2 %; 512-byte NALs: 3.7 %, zero-heavy data: 2-3 %), against 0.30-0.34 for the LDS-image kernel; beyond,
                                           its tiles are walked chunk by chunk (0.12 at 384-byte NALs).  1 in 26 = 3.85 % leaves the spread of a
                                           tile's count below the limit.  (Round 2: 1 in 40 of the chunks with a zero PAIR, i.e. ~6 x fewer elements.) */
constexpr int kExactFlagMin = 2;        /* rows of 1 KiB with more flagged chunks than this are asked again, exactly (chunk_pattern_any_dev): one or two
                                           are a start code, most likely, and the second test would buy nothing */
/* Without an arena (round 4): the streaming index-only kernel keeps its pace to about twice that density -- a 2 GiB stream of
 * 448-byte NALs (one chunk in 24 an element) 0.66 ms against the LDS-image kernel's 1.48, of 384-byte NALs 1.13 against 1.50, of
 * 256-byte NALs (1 in 14) 1.47 against 1.57, of 128-byte NALs 2.5 against 1.8 (scripts/pin_time.py) -- so its calls ask the
 * probe with a threshold of their own. */
constexpr uint32_t kDenseOneInIndexOnly = 15;
HBS_HD bool probe_says_dense(uint32_t chunks, uint32_t flagged, uint32_t one_in = kDenseOneIn) { return (uint64_t)flagged * one_in > (uint64_t)chunks; }
enum : int { kGateNone = 0, kGateIfSparse = 1, kGateIfDense = 2, kGateIfSparseIdx = 3, kGateIfDenseIdx = 4 };   /* ...Idx: an index-only call's threshold */
#ifdef __HIPCC__
/* the density probe's verdict (hbs_common.h), by a whole wavefront: lane l reads slot l */
__device__ __forceinline__ bool probe_dense_dev(const RunHeader* __restrict__ hdr, uint32_t one_in = kDenseOneIn)
{
    const int lane = threadIdx.x & 63;
    uint32_t c = hdr->probe_slot[lane][0], f = hdr->probe_slot[lane][1];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { c += __shfl_xor(c, d, 64); f += __shfl_xor(f, d, 64); }
    return probe_says_dense(c, f, one_in);
}
/* a kernel of the automatic mode launched with `gate`: true = the probe rules it out, it returns at once */
__device__ __forceinline__ bool gate_closed(int gate, const RunHeader* __restrict__ hdr)
{
    if (gate == kGateNone) return false;
    const bool idx = gate == kGateIfSparseIdx || gate == kGateIfDenseIdx;
    const bool dense = probe_dense_dev(hdr, idx ? kDenseOneInIndexOnly : kDenseOneIn);
    return (gate == kGateIfSparse || gate == kGateIfSparseIdx) ? dense : !dense;
}
#endif


HBS_HD uint32_t popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
HBS_HD uint32_t ctz64(uint64_t v)  { return (uint32_t)__builtin_ctzll(v); }
/* bits [0, n) set; n in [0, 64] */
HBS_HD uint64_t below(uint32_t n)  { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

/* bytes [n, n+4) of the 8-byte little-endian value {hi,lo}; n in 0..3 */
HBS_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8u * n));
#endif
}

/* 0x80 in every byte of v that is zero (exact, no cross-byte carries) */
HBS_HD uint32_t zero_bytes(uint32_t v)
{
    uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | v | 0x7F7F7F7Fu);
}

/* gather the four 0x80 marks of m into bits 0..3 */
HBS_HD uint32_t movemask4(uint32_t m)
{
    return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;
}

/* SplitMix64 finaliser: the synthetic-stream PRNG */
HBS_HD uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kSalt   = 0xD1B54A32D192ED03ull;

} // namespace hbs
#endif
