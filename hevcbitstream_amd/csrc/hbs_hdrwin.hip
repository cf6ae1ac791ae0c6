/*
 * hbs_hdrwin.hip -- header windows: what hbs_index_parse puts between the index-only scan and the header parse so that
 * BASELINE config 3 ("NAL index + VPS / SPS / PPS / slice_segment_header parse") does not pay for an RBSP arena.
 *
 * The reference strips a whole NAL before it reads a header of < 100 bytes from it (read_hevc_nal_unit:
 * nal_to_rbsp, hevc_stream.c:161-179, then the bit reader over the RBSP).  hbs_index_extract + hbs_parse_headers do the
 * same in bulk: 2 bytes of traffic per stream byte for the arena.  Here only the part of each NAL the parse can look
 * at is stripped, into a small arena of its own:
 *     slice segments            the first `window` RBSP bytes (slot k x window)
 *     VPS / SPS / PPS           the whole NAL (they are read to their end: more_rbsp_data looks for the last 1 bit),
 *                               bump-allocated behind the slots
 *     everything else           16 bytes (read_hevc_nal_unit returns -1 for them behind the 2-byte NAL header, :221)
 * with the same rule K12's arena follows (every byte but a 03 behind two zeros, h264_nal.c:160-176), and the parse
 * (K4, unchanged) runs on a copy of the index that points into it, each NAL's length cut to what was stripped.
 * A slice header that ends at least 8 bytes inside its window was read from the same bytes in the same order as from a full
 * RBSP (the bit reader's cursor only moves forward, and it never reads at or behind its `size`), so every field is what
 * hbs_parse_headers gives -- except slice_data_size, which counts the bytes behind the header and is put right afterwards
 * (k_hdr_fix).  A header that comes closer to the end of its window than that is REPORTED, never guessed:
 * HBS_E_CAPACITY in the parse summary, rc of that NAL = INT32_MIN; the caller asks again with a larger window (or takes
 * the arena path).  512 bytes hold any slice header without hundreds of entry points.
 */
#include <hip/hip_runtime.h>
#include <climits>
#include "hbs_common.h"
#include "hbs_hdrwin.h"

namespace hbs {

namespace {

constexpr uint32_t kPsetCap = 1u << 16;     /* a parameter set longer than this is reported like a window that is too small */
constexpr uint32_t kOtherBytes = 16;

__device__ __forceinline__ bool is_pset(int t) { return t >= 32 && t <= 34; }

/* One wavefront per NAL, 512 raw bytes a step (8 per lane): a slice's window is one step, two when it holds emulation
 * prevention bytes.  The first kEpbNote of them are noted -- as the number of RBSP bytes in front of each -- so that k_hdr_fix
 * can turn an RBSP offset inside the window into a stream offset without walking the bytes again. */
constexpr int kEpbNote = 6;
struct EpbNote { uint16_t n; uint16_t at[kEpbNote]; uint16_t pad; };        /* 16 bytes per NAL */

__global__ __launch_bounds__(256)
void k_hdr_strip(const uint8_t* __restrict__ stream, const hbs_nal_entry* __restrict__ index, uint64_t nals, uint32_t window,
                 uint8_t* __restrict__ arena, uint64_t slots_bytes, uint64_t arena_bytes, unsigned long long* __restrict__ bump,
                 hbs_nal_entry* __restrict__ idx2, EpbNote* __restrict__ notes, uint32_t* __restrict__ flags)
{
    struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4u;
    for (uint64_t k = wave; k < nals; k += nwaves) {
        const hbs_nal_entry e = index[k];
        const uint64_t raw = e.end - e.start;
        /* this lane's 8 bytes of the first step, asked for before anything else (the type is in the first of them) */
        const uint8_t* const src = stream + e.start;
        uint64_t v = ~0ull;
        {
            const uint64_t j = 8ull * (uint64_t)lane;
            if (j + 8 <= raw) v = reinterpret_cast<const U8*>(src + j)->v;
            else for (uint32_t t = 0; t < 8 && j + t < raw; ++t) v = (v & ~(0xFFull << (8 * t))) | ((uint64_t)src[j + t] << (8 * t));
        }
        const int type = raw > 0 ? (int)(((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) >> 1) & 0x3Fu) : 63;
        uint32_t want = type < 32 ? window : is_pset(type) ? kPsetCap : kOtherBytes;
        if (want > e.rbsp_len) want = e.rbsp_len;
        uint64_t dst_off = k * (uint64_t)window;
        if (is_pset(type)) {
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(bump, (unsigned long long)((want + 15u) & ~15u));
            dst_off = slots_bytes + (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got));
            if (dst_off + want > arena_bytes) {                  /* the bump area is full: reported, nothing written */
                if (lane == 0) atomicOr(flags, 1u);
                dst_off = 0; want = 0;
            }
        }
        uint8_t* const dst = arena + dst_off;
        uint32_t out = 0, tail2 = 0xFFFFu;                       /* the two bytes in front of this step's first (low byte = the nearer) */
        uint32_t noted = 0;
        EpbNote note;
        note.n = 0; note.pad = 0;
#pragma unroll
        for (int i = 0; i < kEpbNote; ++i) note.at[i] = 0xFFFFu;
        for (uint64_t j0 = 0; j0 < raw && out < want; j0 += 512) {
            if (j0) {
                const uint64_t j = j0 + 8ull * (uint64_t)lane;
                v = ~0ull;
                if (j + 8 <= raw) v = reinterpret_cast<const U8*>(src + j)->v;
                else for (uint32_t t = 0; t < 8 && j + t < raw; ++t) v = (v & ~(0xFFull << (8 * t))) | ((uint64_t)src[j + t] << (8 * t));
            }
            /* the two bytes in front of my eight: the previous lane's last two, or what the previous step left */
            const uint32_t hi = (uint32_t)(v >> 48);                                         /* my bytes 6, 7 */
            uint32_t prev = (uint32_t)__shfl_up((int)hi, 1, 64);
            if (lane == 0) prev = ((tail2 & 0xFFu) << 8) | (tail2 >> 8);                     /* byte -2 low, byte -1 high: as bytes 6, 7 */
            /* x = bytes -2, -1, 0 .. 7 as a 10-byte little-endian number */
            const uint64_t lo10 = (v << 16) | (uint64_t)(prev & 0xFFFFu);
            const uint32_t top2 = (uint32_t)(v >> 48);
            uint32_t keep = 0, epbs = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint64_t w = t < 6 ? (lo10 >> (8 * t)) & 0xFFFFFFull
                                         : ((lo10 >> (8 * t)) | ((uint64_t)top2 << (64 - 8 * t))) & 0xFFFFFFull;   /* bytes t-2, t-1, t */
                const bool in = j0 + 8ull * (uint64_t)lane + (uint64_t)t < raw;
                const bool epb = in && w == 0x030000ull;
                if (in && !epb) keep |= 1u << t;
                if (epb) epbs |= 1u << t;
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(keep);
            uint32_t inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t2 = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += t2; }
            uint32_t pos = out + inc - cnt;
            /* rare: note where (as RBSP bytes in front of it), lanes in order */
            uint64_t em = __ballot(epbs != 0u);
            while (em) {
                const int l = (int)__builtin_ctzll(em);
                em &= em - 1;
                const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)epbs, l), kp = (uint32_t)__builtin_amdgcn_readlane((int)keep, l);
                const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane((int)pos, l);
                for (uint32_t m = eb; m; m &= m - 1) {
                    const uint32_t t = (uint32_t)__builtin_ctz(m);
                    const uint32_t at = ps + (uint32_t)__builtin_popcount(kp & ((1u << t) - 1u));
#pragma unroll
                    for (int i = 0; i < kEpbNote; ++i) if ((uint32_t)i == noted && at <= 0xFFFEu) note.at[i] = (uint16_t)at;     /* (no dynamic index: registers) */
                    ++noted;
                }
            }
            if (__ballot(epbs != 0u) == 0ull) {
                /* no emulation prevention byte in this step (nearly always): a lane's bytes are consecutive in the window -- one
                 * 8-byte store where all eight are kept and fit (round 5; byte by byte this kernel took 63 us for the 100 k windows
                 * of BASELINE config 3) */
                if (keep == 0xFFu && pos + 8u <= want) reinterpret_cast<U8*>(dst + pos)->v = v;
                else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) if ((keep >> t) & 1u) { if (pos < want) dst[pos] = (uint8_t)(v >> (8 * t)); ++pos; }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if ((keep >> t) & 1u) {
                        if (pos < want) dst[pos] = (uint8_t)(v >> (8 * t));
                        ++pos;
                    }
                }
            }
            out += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
            tail2 = ((last & 0xFFu) << 8) | (last >> 8);         /* byte 7 low (the nearer), byte 6 high */
        }
        if (out > want) out = want;
        if (lane == 0) {
            hbs_nal_entry o = e;
            o.rbsp_off = dst_off;
            o.rbsp_len = out;                                    /* = min(rbsp_len, want): the bit reader's `size` */
            idx2[k] = o;
            note.n = (uint16_t)(noted > 0xFFFFu ? 0xFFFFu : noted);
            notes[k] = note;
        }
    }
}

/* behind the parse: slice_data_size counts to the end of the REAL RBSP; a header that came within 8 bytes of the end of a
 * window that does not hold the whole NAL is reported; the payload's place in the stream */
__global__ __launch_bounds__(256)
void k_hdr_fix(const hbs_nal_entry* __restrict__ index, const hbs_nal_entry* __restrict__ idx2, uint64_t nals,
               ParsedWin* __restrict__ parsed, const uint32_t* __restrict__ flags, hbs_summary* __restrict__ summary,
               const uint8_t* __restrict__ stream, const EpbNote* __restrict__ notes, unsigned long long* __restrict__ payload_off, int compact)
{
    bool overflow = false;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nals; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t full = index[k].rbsp_len, have = idx2[k].rbsp_len;
        ParsedWin p = parsed[k];
        const bool slice = p.nal_unit_type >= 0 && p.nal_unit_type < 32 && (p.struct_off != ~0ull || compact);    /* (a compact parse gives slices no struct) */
        const bool pset = p.nal_unit_type >= 32 && p.nal_unit_type <= 34;
        if (payload_off) payload_off[k] = ~0ull;
        if (have < full) {
            if (pset || (slice && p.slice_data_off + 8u > have)) {
                overflow = true;
                p.rc = INT_MIN;
                parsed[k] = p;
                continue;
            }
            if (slice && p.slice_data_off != 0u) {
                p.slice_data_size += (int32_t)(full - have);
                parsed[k] = p;
            }
        }
        if (payload_off && slice && p.slice_data_off != 0u && p.slice_data_off <= full) {
            /* RBSP byte slice_data_off in the stream: its index plus the emulation prevention bytes in front of it */
            const EpbNote nt = notes[k];
            const uint64_t s = index[k].start;
            if (p.slice_data_off <= have && nt.n <= (uint16_t)kEpbNote) {
                uint32_t e = 0;
#pragma unroll
                for (int i = 0; i < kEpbNote; ++i) e += (i < (int)nt.n && (uint32_t)nt.at[i] <= p.slice_data_off) ? 1u : 0u;
                payload_off[k] = s + p.slice_data_off + e;
            } else {                                             /* more of them than were noted, or behind the window: walk */
                const uint64_t raw = index[k].end - s;
                uint32_t kept = 0, z = 0;
                uint64_t j = 0;
                for (; j < raw && kept < p.slice_data_off; ++j) {
                    const uint32_t b = stream[s + j];
                    if (b == 3u && z >= 2u) { z = 0; continue; }
                    z = b == 0u ? z + 1u : 0u;
                    ++kept;
                }
                if (j < raw && stream[s + j] == 3u && z >= 2u) ++j;
                payload_off[k] = s + j;
            }
        }
    }
    if (__syncthreads_or(overflow ? 1 : 0) && threadIdx.x == 0) summary->error = HBS_E_CAPACITY;
    if (blockIdx.x == 0 && threadIdx.x == 0 && flags[0] != 0u) summary->error = HBS_E_CAPACITY;
}

} // namespace

uint64_t hdrwin_arena_bytes(uint64_t index_cap, uint32_t window, uint64_t stream_bytes)
{
    uint64_t bump = stream_bytes / 64;
    if (bump < (8ull << 20)) bump = 8ull << 20;
    return index_cap * (uint64_t)window + bump;
}

hipError_t launch_hdr_strip(const HdrWinArgs& a, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(a.bump, 0, 16, st);           /* bump counter and flags */
    if (e != hipSuccess) return e;
    if (a.nals) {
        uint64_t blocks = (a.nals + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        k_hdr_strip<<<dim3((unsigned)blocks), 256, 0, st>>>(a.stream, a.index, a.nals, a.window, a.arena, a.index_cap * (uint64_t)a.window,
                                                             a.arena_bytes, a.bump, a.idx2, static_cast<EpbNote*>(a.notes), reinterpret_cast<uint32_t*>(a.bump + 1));
    }
    return hipGetLastError();
}

hipError_t launch_hdr_fix(const HdrWinArgs& a, void* parsed, hbs_summary* summary, unsigned long long* payload_off, hipStream_t st, int compact)
{
    if (a.nals) {
        uint64_t blocks = (a.nals + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        k_hdr_fix<<<dim3((unsigned)blocks), 256, 0, st>>>(a.index, a.idx2, a.nals, static_cast<ParsedWin*>(parsed),
                                                           reinterpret_cast<const uint32_t*>(a.bump + 1), summary, a.stream, static_cast<const EpbNote*>(a.notes), payload_off, compact);
    }
    return hipGetLastError();
}

} // namespace hbs
