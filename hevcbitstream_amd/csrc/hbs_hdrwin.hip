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

/* one wavefront per NAL, 64 raw bytes a step */
__global__ __launch_bounds__(256)
void k_hdr_strip(const uint8_t* __restrict__ stream, const hbs_nal_entry* __restrict__ index, uint64_t nals, uint32_t window,
                 uint8_t* __restrict__ arena, uint64_t slots_bytes, uint64_t arena_bytes, unsigned long long* __restrict__ bump,
                 hbs_nal_entry* __restrict__ idx2, uint32_t* __restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4u;
    for (uint64_t k = wave; k < nals; k += nwaves) {
        const hbs_nal_entry e = index[k];
        const uint64_t raw = e.end - e.start;
        const int type = raw > 0 ? (int)((stream[e.start] >> 1) & 0x3Fu) : 63;
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
        uint32_t out = 0, p1 = 0xFF, p2 = 0xFF;                  /* the two bytes in front of this step's first */
        for (uint64_t j0 = 0; j0 < raw && out < want; j0 += 64) {
            const uint64_t j = j0 + (uint64_t)lane;
            const uint32_t b = j < raw ? stream[e.start + j] : 0xFFu;
            uint32_t q1 = (uint32_t)__shfl_up((int)b, 1, 64), q2 = (uint32_t)__shfl_up((int)b, 2, 64);
            if (lane == 0) { q1 = p1; q2 = p2; }
            if (lane == 1) q2 = p1;
            const bool keep = j < raw && !(b == 3u && q1 == 0u && q2 == 0u);
            const uint64_t m = __ballot(keep);
            const uint32_t rank = out + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep && rank < want) dst[rank] = (uint8_t)b;
            out += (uint32_t)__builtin_popcountll(m);
            p2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 62);
            p1 = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
        }
        if (out > want) out = want;
        if (lane == 0) {
            hbs_nal_entry o = e;
            o.rbsp_off = dst_off;
            o.rbsp_len = out;                                    /* = min(rbsp_len, want): the bit reader's `size` */
            idx2[k] = o;
        }
    }
}

/* behind the parse: slice_data_size counts to the end of the REAL RBSP; a header that came within 8 bytes of the end of a
 * window that does not hold the whole NAL is reported */
__global__ __launch_bounds__(256)
void k_hdr_fix(const hbs_nal_entry* __restrict__ index, const hbs_nal_entry* __restrict__ idx2, uint64_t nals,
               ParsedWin* __restrict__ parsed, const uint32_t* __restrict__ flags, hbs_summary* __restrict__ summary,
               const uint8_t* __restrict__ stream, unsigned long long* __restrict__ payload_off)
{
    bool overflow = false;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nals; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t full = index[k].rbsp_len, have = idx2[k].rbsp_len;
        ParsedWin p = parsed[k];
        const bool slice = p.nal_unit_type >= 0 && p.nal_unit_type < 32 && p.struct_off != ~0ull;
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
            /* the stream offset of RBSP byte slice_data_off: walk the NAL's first bytes once more (a few hundred at most) */
            const uint64_t s = index[k].start, raw = index[k].end - s;
            uint32_t kept = 0, z = 0;
            uint64_t j = 0;
            for (; j < raw && kept < p.slice_data_off; ++j) {
                const uint32_t b = stream[s + j];
                if (b == 3u && z >= 2u) { z = 0; continue; }
                z = b == 0u ? z + 1u : 0u;
                ++kept;
            }
            /* an emulation prevention byte right in front of the payload's first byte belongs to the gap */
            if (j < raw && stream[s + j] == 3u && z >= 2u) ++j;
            payload_off[k] = s + j;
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
                                                             a.arena_bytes, a.bump, a.idx2, reinterpret_cast<uint32_t*>(a.bump + 1));
    }
    return hipGetLastError();
}

hipError_t launch_hdr_fix(const HdrWinArgs& a, void* parsed, hbs_summary* summary, unsigned long long* payload_off, hipStream_t st)
{
    if (a.nals) {
        uint64_t blocks = (a.nals + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        k_hdr_fix<<<dim3((unsigned)blocks), 256, 0, st>>>(a.index, a.idx2, a.nals, static_cast<ParsedWin*>(parsed),
                                                           reinterpret_cast<const uint32_t*>(a.bump + 1), summary, a.stream, payload_off);
    }
    return hipGetLastError();
}

} // namespace hbs
