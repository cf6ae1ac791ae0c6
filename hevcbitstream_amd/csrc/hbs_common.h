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
#define HBS_DENSE_ONE_IN 44
#endif
constexpr uint32_t kDenseOneIn = HBS_DENSE_ONE_IN;    /* Round 3: the probe counts ELEMENTS (chunks a pattern 00 00 {<=3} ends in, neighbours seen).  Round 6: beyond one
                                           element in 44 chunks the event-sparse kernel's 24-row geometry takes over from the 48-row one (2 GiB sweep, profiles/r06:
                                           NALs of 1 KiB -- one in 53 -- 0.58 of peak on 48 rows against 0.55 on 24; 768 bytes -- one in 40 -- 0.535 / 0.530;
                                           512 bytes 0.485 / 0.516; the zero-heavy stress stream -- one in 41 -- 0.524 / 0.561), where until round 5 the 48-row
                                           geometry ran up to one in 26 and the LDS-image kernel took the rest at 0.25-0.31. */
constexpr int kExactFlagMin = 2;        /* rows of 1 KiB with more flagged chunks than this are asked again, exactly (chunk_pattern_any_dev): one or two
                                           are a start code, most likely, and the second test would buy nothing */
/* Without an arena: the streaming index-only kernel (hbs_scan5.hip) records eight elements a KiB since round 6 and keeps its pace to
 * one chunk in 9 an element -- 2 GiB of 384 / 320 / 256 / 192-byte NALs 0.45 / 0.42 / 0.39 / 0.33 of peak (read), where the 24-row
 * geometry without an arena reaches 0.34 / 0.32 / 0.32 / 0.29 and the LDS-image kernel 0.21; at 128 bytes (one in 6.7) its tiles pass
 * the record space and are streamed twice -- so its calls ask the probe with a threshold of their own. */
constexpr uint32_t kDenseOneInIndexOnly = 9;
HBS_HD bool probe_says_dense(uint32_t chunks, uint32_t flagged, uint32_t one_in = kDenseOneIn) { return (uint64_t)flagged * one_in > (uint64_t)chunks; }
/* Round 6: three classes.  Between "sparse" (above) and "dense" lies what the 24-row geometry of the event-sparse kernel is for
 * (hbs_scan4_r24.hip: 1024 elements per 96 KiB tile = one chunk in 6): streams of NALs of ~120 to ~450 bytes.  A 2 GiB sweep
 * (profiles/r06): extract 0.32 / 0.38 / 0.43 / 0.45 of peak at 128 / 192 / 256 / 384-byte NALs against the LDS-image kernel's
 * 0.25 / 0.27 / 0.29 / 0.31, and 0.15 against 0.23 at 64 bytes, where its tiles pass the limit and are walked by rows -- so
 * "dense" begins at 2 elements in kMidTwoIn chunks (one in 6.5; 128-byte NALs: one in 6.7).  Index-only calls of 1 GiB and more
 * (the streaming kernel's) have no middle class: beyond one element in 9 chunks the 24-row geometry without an arena is 0.25 against
 * the LDS-image kernel's 0.22 at 128 bytes, and the launch that rules itself out on every other stream cost 2 % of a 1 GiB call. */
constexpr uint32_t kMidTwoIn = 13;
enum : int { kProbeSparse = 0, kProbeMid = 1, kProbeDense = 2 };
HBS_HD int probe_class(uint32_t chunks, uint32_t flagged, bool index_only)
{
    if (!probe_says_dense(chunks, flagged, index_only ? kDenseOneInIndexOnly : kDenseOneIn)) return kProbeSparse;
    return (uint64_t)flagged * kMidTwoIn > 2ull * (uint64_t)chunks ? kProbeDense : kProbeMid;
}
/* the kernel an automatic call runs: 4 event-sparse (48 rows), 5 streaming index-only, 6 event-sparse with 24 rows, 2 LDS image */
HBS_HD int probe_variant(uint32_t chunks, uint32_t flagged, bool index_only)
{
    const int c = probe_class(chunks, flagged, index_only);
    if (index_only) return c == kProbeSparse ? 5 : 2;      /* (no 24-row geometry behind the streaming kernel: see launch_scan_extract) */
    return c == kProbeSparse ? 4 : c == kProbeMid ? 6 : 2;
}
enum : int { kGateNone = 0, kGateIfSparse = 1, kGateIfDense = 2, kGateIfSparseIdx = 3, kGateIfDenseIdx = 4, kGateIfMid = 5 };   /* ...Idx: an index-only call's threshold (sparse or not) */
#ifdef __HIPCC__
/* the density probe's verdict, by a whole wavefront: lane l reads slot l */
__device__ __forceinline__ int probe_class_dev(const RunHeader* __restrict__ hdr, bool index_only)
{
    const int lane = threadIdx.x & 63;
    uint32_t c = hdr->probe_slot[lane][0], f = hdr->probe_slot[lane][1];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { c += __shfl_xor(c, d, 64); f += __shfl_xor(f, d, 64); }
    return probe_class(c, f, index_only);
}
/* a kernel of the automatic mode launched with `gate`: true = the probe rules it out, it returns at once */
__device__ __forceinline__ bool gate_closed(int gate, const RunHeader* __restrict__ hdr)
{
    if (gate == kGateNone) return false;
    const bool idx = gate == kGateIfSparseIdx || gate == kGateIfDenseIdx;
    const int c = probe_class_dev(hdr, idx);
    if (idx) return (gate == kGateIfSparseIdx) != (c == kProbeSparse);       /* two ways only: the streaming kernel, or the LDS image */
    const int want = gate == kGateIfSparse ? kProbeSparse : gate == kGateIfMid ? kProbeMid : kProbeDense;
    return c != want;
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
