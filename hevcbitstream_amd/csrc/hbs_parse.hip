/*
 * hbs_parse.hip -- K4 / K5 drivers: header parse and header writers, one NAL per
 * lane, over the RBSP arena that K12 produced (see hbs_parse.h for the syntax walk).
 *
 *   k4_plan    per NAL: type from RBSP bytes 0-1 (reference hevc_stream.c:176-179),
 *              bytes of the struct it parses into
 *   k4_scan_*  struct arena offsets; for every NAL the ordinal of
 *              the last SPS / PPS in front of it (the reference's "h->sps / h->pps
 *              as left by the last parse", hevc_stream.c:800-801)
 *   k4_parse   a wavefront walks 64 NALs in lock step; launch 1 parameter sets (its spare
 *              workgroups clear the slice slots), launch 2 slices against them
 *   k4_small   all of the above for at most 64 NALs, in one launch of one wavefront
 *   k5_write   the same walk in write mode (structs -> RBSP)
 */
#include <hip/hip_runtime.h>
#include "hbs_parse.h"
#include "hbs_parse_compact.h"
#include "hbs_parse_fix.h"
#include "hbs_parse_ext.h"
#include "hbs_parse_launch.h"

namespace hbs {


/* compact: slices get no slot in the struct arena (k4_want gives the listed ones theirs back) and every NAL a zero record */
__global__ void k4_plan(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n,
                        ParsedNal* __restrict__ parsed, unsigned long long* __restrict__ slot_size, uint32_t* __restrict__ deps,
                        SliceCompact* __restrict__ compact = nullptr, unsigned long long* __restrict__ total = nullptr, uint32_t* __restrict__ err = nullptr,
                        uint32_t* __restrict__ div_flag = nullptr, uint32_t* __restrict__ fix_count = nullptr)
{
    if (compact && blockIdx.x == 0 && threadIdx.x == 0) { *total = 0ull; *err = 0u; *div_flag = 0u; fix_count[0] = 0u; fix_count[1] = 0u; }
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const hbs_nal_entry e = idx[k];
        ParsedNal p;
        p.rc = -1; p.nal_unit_type = -1; p.nal_layer_id = -1; p.nal_temporal_id_plus1 = -1;
        p.struct_off = ~0ull; p.slice_data_size = 0; p.slice_data_off = 0;
        uint64_t sz = 0;
        if (!(e.status & HBS_ST_ERROR)) {            /* nal_to_rbsp failed: read_hevc_nal_unit returns before the header (:167) */
            nal_header_of(rbsp + e.rbsp_off, e.rbsp_len, p);
            sz = slot_bytes_of(p.nal_unit_type);
            if (compact && is_slice_type_nal(p.nal_unit_type)) sz = 0;
        }
        parsed[k] = p;
        slot_size[k] = sz;
        deps[k] = 0u;
        if (compact) compact[k] = compact_zero();
    }
}

/* hbs_parse_materialize: the listed NALs, where they are slices, get a slot after all */
__global__ void k4_want(const ParsedNal* __restrict__ parsed, uint64_t n, const uint64_t* __restrict__ list, uint64_t m,
                        unsigned long long* __restrict__ slot_size)
{
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = list[j];
        if (k < n && is_slice_type_nal(parsed[k].nal_unit_type)) slot_size[k] = slot_bytes_of(parsed[k].nal_unit_type);
    }
}

/* the NAL types the reference has readers for but never dispatches (hbs_parse_ext.h), one NAL per thread:
 * they are a few bytes each and there are a handful per picture */
__global__ void k4_ext(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n,
                       ParsedNal* __restrict__ parsed, hbs_ext_nal* __restrict__ ext)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const hbs_nal_entry e = idx[k];
        hbs_ext_nal x;
        ParsedNal p;
        p.nal_unit_type = -1;
        if (!(e.status & HBS_ST_ERROR)) nal_header_of(rbsp + e.rbsp_off, e.rbsp_len, p);
        if (is_extended_nal_type(p.nal_unit_type)) {
            const int consumed = (int)(e.end - e.start) - ((e.status & HBS_ST_TRAILING03) ? 1 : 0);
            const int32_t rc = read_extended_nal(rbsp + e.rbsp_off, e.rbsp_len, p.nal_unit_type, consumed, &x);
            parsed[k].rc = rc;
        } else {
            x.num_sei_messages = 0; x.primary_pic_type = 0; x.filler_bytes = 0; x.reserved = 0;
            for (int i = 0; i < HBS_SEI_MAX_MESSAGES; ++i) { x.sei[i].payloadType = 0; x.sei[i].payloadSize = 0; x.sei[i].payload_off = 0; x.sei[i].reserved = 0; }
        }
        ext[k] = x;
    }
}

hipError_t launch_parse_extended(const uint8_t* rbsp, const hbs_nal_entry* index, uint64_t n, ParsedNal* parsed, hbs_ext_nal* ext, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    k4_ext<<<dim3((unsigned)blocks), 256, 0, st>>>(rbsp, index, n, parsed, ext);
    return hipGetLastError();
}

/* ---- exclusive sum of slot sizes, and last SPS / PPS ordinal before each NAL ----------------
 * three small launches: per-block (sum, last SPS, last PPS) of contiguous slices, a scan of the
 * 1024 block results, and the slices again with their carry-in (coalesced 256-wide chunks). */
constexpr int kScan4Blocks = 1024;

struct Scan3 { unsigned long long sum; long long sps, pps; };      /* sps/pps: ordinal of the last one seen, -1 = none */

__device__ __forceinline__ Scan3 scan3_join(const Scan3& a, const Scan3& b)     /* a in front of b */
{
    Scan3 r;
    r.sum = a.sum + b.sum;
    r.sps = b.sps > a.sps ? b.sps : a.sps;
    r.pps = b.pps > a.pps ? b.pps : a.pps;
    return r;
}
__device__ __forceinline__ Scan3 scan3_shfl_up(const Scan3& x, int d)
{
    Scan3 t;
    t.sum = __shfl_up(x.sum, d, 64); t.sps = __shfl_up(x.sps, d, 64); t.pps = __shfl_up(x.pps, d, 64);
    return t;
}

/* across the 256 threads of a workgroup: returns the join of everything in front of this thread
 * (exclusive), *total = join of all (same in every thread) */
__device__ __forceinline__ Scan3 block_excl_scan3(const Scan3& x, Scan3* wsum /* [4] LDS */, Scan3& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    Scan3 inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Scan3 t = scan3_shfl_up(inc, d);
        if (lane >= d) inc = scan3_join(t, inc);
    }
    Scan3 ex = scan3_shfl_up(inc, 1);
    if (lane == 0) { ex.sum = 0; ex.sps = -1; ex.pps = -1; }
    __syncthreads();                       /* wsum may still be read from the previous round */
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    Scan3 before, tot;
    before.sum = 0; before.sps = -1; before.pps = -1;
    tot = before;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wv) before = scan3_join(before, wsum[w]); tot = scan3_join(tot, wsum[w]); }
    total = tot;
    return scan3_join(before, ex);
}

__device__ __forceinline__ Scan3 scan3_of(const ParsedNal* parsed, const unsigned long long* slot_size, uint64_t i, bool valid)
{
    Scan3 x;
    x.sum = 0; x.sps = -1; x.pps = -1;
    if (valid) {
        x.sum = slot_size[i];
        const int t = parsed[i].nal_unit_type;
        if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) x.sps = (long long)i;
        if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT) x.pps = (long long)i;
    }
    return x;
}

__global__ __launch_bounds__(256)
void k4_scan_reduce(const ParsedNal* __restrict__ parsed, const unsigned long long* __restrict__ slot_size, uint64_t n,
                    Scan3* __restrict__ part)
{
    __shared__ Scan3 wsum[4];
    const uint64_t per = (n + kScan4Blocks - 1) / kScan4Blocks;
    const uint64_t lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    Scan3 acc;
    acc.sum = 0; acc.sps = -1; acc.pps = -1;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) acc = scan3_join(acc, scan3_of(parsed, slot_size, i, true));
    Scan3 tot;
    (void)block_excl_scan3(acc, wsum, tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScan4Blocks)
void k4_scan_parts(Scan3* __restrict__ part, unsigned long long* __restrict__ total)
{
    __shared__ unsigned long long ssum[kScan4Blocks];
    __shared__ long long ssps[kScan4Blocks], spps[kScan4Blocks];
    const int tid = threadIdx.x;
    const Scan3 mine = part[tid];
    ssum[tid] = mine.sum; ssps[tid] = mine.sps; spps[tid] = mine.pps;
    __syncthreads();
    for (int d = 1; d < kScan4Blocks; d <<= 1) {
        unsigned long long t = 0; long long a = -1, b = -1;
        if (tid >= d) { t = ssum[tid - d]; a = ssps[tid - d]; b = spps[tid - d]; }
        __syncthreads();
        ssum[tid] += t;
        if (a > ssps[tid]) ssps[tid] = a;
        if (b > spps[tid]) spps[tid] = b;
        __syncthreads();
    }
    Scan3 ex;                                   /* everything in front of block tid */
    ex.sum = ssum[tid] - mine.sum;
    ex.sps = tid ? ssps[tid - 1] : -1;
    ex.pps = tid ? spps[tid - 1] : -1;
    part[tid] = ex;
    if (tid == kScan4Blocks - 1) *total = ssum[tid];
}

template <bool kAssignSlots>
__global__ __launch_bounds__(256)
void k4_scan_apply(ParsedNal* __restrict__ parsed, const unsigned long long* __restrict__ slot_size, uint64_t n,
                   const Scan3* __restrict__ part, long long* __restrict__ ctx_sps, long long* __restrict__ ctx_pps)
{
    __shared__ Scan3 wsum[4];
    const uint64_t per = (n + kScan4Blocks - 1) / kScan4Blocks;
    const uint64_t lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    Scan3 carry = part[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const Scan3 x = scan3_of(parsed, slot_size, i, i < hi);
        Scan3 tot;
        const Scan3 ex = scan3_join(carry, block_excl_scan3(x, wsum, tot));
        if (i < hi) {
            if (kAssignSlots) parsed[i].struct_off = x.sum ? ex.sum : ~0ull;
            ctx_sps[i] = ex.sps; ctx_pps[i] = ex.pps;
        }
        carry = scan3_join(carry, tot);
    }
}

/* the memset of every struct a parse fills (hevc_stream.c:250, :310, :425, init_slice_hevc :19-24).
 * Slice slots are cleared by spare workgroups of the parameter-set launch (the parameter sets
 * themselves are few and their parse is one long dependent chain: the clearing hides behind it);
 * a parameter set's slot is cleared by the wavefront that parses it. */
__device__ __forceinline__ void zero_slot(uint8_t* __restrict__ dst, uint64_t bytes, int lane)
{
    uint4* q = reinterpret_cast<uint4*>(dst);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)lane; i < bytes / 16; i += 64) q[i] = z;       /* slots are multiples of 16 */
}

__device__ __forceinline__ void zero_slice_slots(const ParsedNal* __restrict__ parsed, uint64_t n, uint8_t* __restrict__ structs,
                                                 uint64_t structs_cap, uint64_t wave, uint64_t nwaves, int lane)
{
    const uint64_t slot = slot_bytes_of(HEVC_NAL_UNIT_TYPE_TRAIL_R);             /* every slice type has the same struct */
    for (uint64_t chunk = wave; chunk * 64 < n; chunk += nwaves) {
        const uint64_t k = chunk * 64 + (uint64_t)lane;
        uint64_t off = ~0ull;
        if (k < n && is_slice_type_nal(parsed[k].nal_unit_type)) off = parsed[k].struct_off;
        if (off != ~0ull && off + slot > structs_cap) off = ~0ull;
        uint64_t todo = __ballot(off != ~0ull);
        while (todo) {                                                           /* wave-uniform */
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint64_t oj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), j) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, j);
            zero_slot(structs + oj, slot, lane);
        }
    }
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kLaneWin = 64;
constexpr uint32_t kLaneWinStride = kLaneWin + 4;

/* the first wb bytes of a lane's RBSP into its LDS window */
__device__ __forceinline__ void stage_window(uint8_t* my_win, const uint8_t* __restrict__ src, uint32_t wb)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4_t v; };
#pragma unroll
    for (uint32_t i = 0; i < kLaneWin; i += 16) {
        if (i + 16 <= wb) {
            const u32x4_t v = reinterpret_cast<const U16*>(src + i)->v;
            uint32_t* d = reinterpret_cast<uint32_t*>(my_win + i);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            for (uint32_t b = i; b < wb; ++b) my_win[b] = src[b];
        }
    }
}

/* one lane, one NAL: the walk over its syntax into the (cleared) struct at dst.  sps_slot = the SPS in force with
 * its derived tables behind it, pps_struct = the PPS in force (nullptr: none, the all-zero set); returns the
 * number of trace records the walk produced */
template <int kMode>
__device__ __forceinline__ uint32_t parse_lane(int type, bool slice, const hbs_nal_entry& e, uint8_t* dst,
                                               const uint8_t* my_win, uint32_t wb, const uint8_t* src,
                                               const uint8_t* sps_slot, const uint8_t* pps_struct, const uint8_t* zeros,
                                               ParsedNal& out, TraceRec* trace, uint32_t trace_cap, RpsRow* own_row,
                                               RpsTables* seq_tables = nullptr, int* diverged = nullptr, uint32_t* deps_out = nullptr)
{
    ParserT<kMode> ps;
    ps.b.win = my_win; ps.b.full = src; ps.b.win_bytes = wb; ps.b.size = e.rbsp_len; ps.b.pos = 16;   /* past the NAL header */
    ps.b.tr = trace; ps.b.tr_cap = trace_cap; ps.b.tr_n = 0; ps.b.wbuf = nullptr;
    ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
    const hevc_sps_t* zero_sps = reinterpret_cast<const hevc_sps_t*>(zeros);
    const hevc_pps_t* zero_pps = reinterpret_cast<const hevc_pps_t*>(zeros);
    const hevc_sps_t* last_sps = zero_sps;
    const hevc_pps_t* last_pps = zero_pps;
    if (slice) {
        reinterpret_cast<hevc_slice_header_t*>(dst)->collocated_from_l0_flag = 1;
        if (sps_slot) {
            last_sps = reinterpret_cast<const hevc_sps_t*>(sps_slot);
            ps.sps_rps = reinterpret_cast<const RpsTables*>(sps_slot + round16(sizeof(hevc_sps_t)));
        }
        if (pps_struct) last_pps = reinterpret_cast<const hevc_pps_t*>(pps_struct);
        ps.own = own_row;
        /* the reference's way (one NAL after the other, hevc_stream.c:26-32): ONE set of tables that every SPS and
         * every slice's own set writes into and reads from, rows nobody rewrote keeping what was there */
        if (seq_tables) { ps.own = nullptr; ps.out_rps = seq_tables; }
    } else if (type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        ps.out_rps = seq_tables ? seq_tables : reinterpret_cast<RpsTables*>(dst + round16(sizeof(hevc_sps_t)));
    }
    const int consumed = (int)(e.end - e.start) - ((e.status & HBS_ST_TRAILING03) ? 1 : 0);
    parse_one_nal(ps, type, dst, consumed, &out, last_pps, last_sps, zero_pps, zero_sps);
    if (diverged) *diverged = ps.diverged;
    if (deps_out && slice) *deps_out = deps_pack(ps.rec_own, ps.rec_ref, ps.rec_read, ps.own_idx);
    return ps.b.tr_n;
}

/* the same for a slice that has no struct: the walk into a sink (hbs_parse_compact.h), its compact record into *compact */
__device__ __forceinline__ void parse_lane_sink(int type, const hbs_nal_entry& e, const uint8_t* my_win, uint32_t wb, const uint8_t* src,
                                                const uint8_t* sps_slot, const uint8_t* pps_struct, const uint8_t* zeros,
                                                ParsedNal& out, SliceCompact* compact, RpsRow* own_row, int* diverged, uint32_t* deps_out)
{
    ParserT<kModeRead> ps;
    ps.b.win = my_win; ps.b.full = src; ps.b.win_bytes = wb; ps.b.size = e.rbsp_len; ps.b.pos = 16;
    ps.b.tr = nullptr; ps.b.tr_cap = 0; ps.b.tr_n = 0; ps.b.wbuf = nullptr;
    ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
    const hevc_sps_t* zero_sps = reinterpret_cast<const hevc_sps_t*>(zeros);
    const hevc_pps_t* zero_pps = reinterpret_cast<const hevc_pps_t*>(zeros);
    const hevc_sps_t* last_sps = zero_sps;
    const hevc_pps_t* last_pps = zero_pps;
    if (sps_slot) {
        last_sps = reinterpret_cast<const hevc_sps_t*>(sps_slot);
        ps.sps_rps = reinterpret_cast<const RpsTables*>(sps_slot + round16(sizeof(hevc_sps_t)));
    }
    if (pps_struct) last_pps = reinterpret_cast<const hevc_pps_t*>(pps_struct);
    ps.own = own_row;
    const int consumed = (int)(e.end - e.start) - ((e.status & HBS_ST_TRAILING03) ? 1 : 0);
    parse_slice_into_sink(ps, type, consumed, &out, compact, last_pps, last_sps, zero_pps, zero_sps);
    *diverged = ps.diverged;
    *deps_out = deps_pack(ps.rec_own, ps.rec_ref, ps.rec_read, ps.own_idx);
}

/* One NAL per LANE: a wavefront walks 64 consecutive NALs at once.  The walk is a long chain of
 * dependent scalar steps (a bit at a time, as bs.h does), so one parser per wavefront left the
 * machine with ~4000 parsers in flight; the slices of a stream mostly take the same path through
 * the syntax, so 64 of them in lock step diverge little.  Each lane stages the first kLaneWin
 * bytes of its RBSP in LDS (stride kLaneWinStride bytes: lanes reading the same offset hit
 * different banks) and owns one RpsRow of a global scratch for its slice's own short-term RPS.
 * pass 0: parameter sets (workgroups [0, parse_blocks)) while the workgroups behind them clear the
 * slice slots; pass 1: slices against them.  kMode: plain parse, or parse + per-field trace (the
 * debug reader's variant of the syntax, see hbs_parse.h). */
#ifndef HBS_PARSE_LANES
#define HBS_PARSE_LANES 64
#endif
constexpr int kParseLanes = HBS_PARSE_LANES;          /* NALs a wavefront walks at once (lanes 0 .. kParseLanes-1) */
constexpr unsigned kZeroBlocks = 1024;                /* spare workgroups of the parameter-set launch that clear slice slots */
/* kSink (pass 1 of a compact parse): the slices WITHOUT a slot, into sinks; otherwise: the NALs with one, as always -- and, in a
 * compact parse, the record of a slice that has a slot (hbs_parse_materialize) taken from its struct */
template <int kMode, bool kSink = false>
__global__ __launch_bounds__(256)
void k4_parse(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int pass,
              ParsedNal* __restrict__ parsed, uint8_t* __restrict__ structs, uint64_t structs_cap,
              const long long* __restrict__ ctx_sps, const long long* __restrict__ ctx_pps,
              const uint8_t* __restrict__ zeros, const uint8_t* __restrict__ init_sps_slot,
              const uint8_t* __restrict__ init_pps, uint32_t* __restrict__ err,
              TraceRec* __restrict__ trace, uint32_t trace_cap, uint32_t* __restrict__ trace_count,
              RpsRow* __restrict__ own_rows /* 64 per wavefront of the grid */, unsigned parse_blocks, uint32_t* __restrict__ div_flag,
              uint32_t* __restrict__ deps, SliceCompact* __restrict__ compact = nullptr, int psets_cleared = 0)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[4][64 * kLaneWinStride];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x >= parse_blocks) {                          /* pass 0 only: the spare workgroups */
        zero_slice_slots(parsed, n, structs, structs_cap, (uint64_t)(blockIdx.x - parse_blocks) * 4u + (uint64_t)wv,
                         (uint64_t)(gridDim.x - parse_blocks) * 4u, lane);
        return;
    }
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (uint64_t)wv;
    const uint64_t nwaves = (uint64_t)parse_blocks * 4u;
    RpsRow* const my_rows = own_rows + wave * 64;
    uint8_t* const my_win = win[wv] + (uint32_t)lane * kLaneWinStride;

    for (uint64_t chunk = wave; chunk * kParseLanes < n; chunk += nwaves) {
        const uint64_t k = chunk * kParseLanes + (uint64_t)lane;
        int type = -1;
        if (k < n && lane < kParseLanes) type = parsed[k].nal_unit_type;
        const bool slice = is_slice_type_nal(type);
        const bool pset = type == HEVC_NAL_UNIT_TYPE_VPS_NUT || type == HEVC_NAL_UNIT_TYPE_SPS_NUT || type == HEVC_NAL_UNIT_TYPE_PPS_NUT;
        /* type < 0: nal_to_rbsp failed, rc stays -1; other types: rc -1, header fields kept (:221) */
        bool active = type >= 0 && (slice || pset) && ((pass == 0) == pset);
        hbs_nal_entry e;
        e.start = e.end = e.rbsp_off = 0; e.rbsp_len = 0; e.status = 0;
        uint64_t off = 0;
        if (active) {
            e = idx[k];
            off = parsed[k].struct_off;
            if (compact && slice && (off == ~0ull) != kSink) {      /* the other slice pass's */
                active = false;
            } else if (kSink) {
                /* no slot: nothing to check */
            } else if (off + slot_bytes_of(type) > structs_cap) {
                atomicMax(err, (uint32_t)(-HBS_E_CAPACITY));
                parsed[k].struct_off = ~0ull;
                active = false;
            }
        }
        if (compact && __ballot(active) == 0ull) continue;     /* (a compact parse runs two slice passes: most wavefronts of one have nothing to do) */
        if (pass == 1) {                                       /* fresh rows for the slices' own short-term RPS, by the whole wave */
            uint4* q = reinterpret_cast<uint4*>(my_rows);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint32_t i = (uint32_t)lane; i < (uint32_t)(64 * sizeof(RpsRow) / 16); i += 64) q[i] = z;
        } else if (!psets_cleared) {                           /* the slots of my parameter sets, by the whole wave (k4_zero_psets did it otherwise) */
            uint64_t todo = __ballot(active);
            while (todo) {
                const int j = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint64_t oj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), j) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, j);
                zero_slot(structs + oj, slot_bytes_of(__builtin_amdgcn_readlane(type, j)), lane);
            }
        }
        /* first bytes of my RBSP into my LDS window */
        const uint8_t* src = rbsp + e.rbsp_off;
        const uint32_t wb = e.rbsp_len < kLaneWin ? e.rbsp_len : kLaneWin;
        if (active) stage_window(my_win, src, wb);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

        if (active) {
            const uint8_t* sps_slot = nullptr;
            const uint8_t* pps_struct = nullptr;
            if (slice) {
                const long long cs = ctx_sps[k], cp = ctx_pps[k];
                if (cs >= 0) { if (parsed[cs].struct_off != ~0ull) sps_slot = structs + parsed[cs].struct_off; }
                else sps_slot = init_sps_slot;                    /* context handed in by the caller */
                if (cp >= 0) { if (parsed[cp].struct_off != ~0ull) pps_struct = structs + parsed[cp].struct_off; }
                else pps_struct = init_pps;
            }
            ParsedNal out = parsed[k];
            int dv = 0;
            uint32_t dp = 0u;
            uint32_t tr_n = 0;
            if (kSink) {
                SliceCompact rec;
                parse_lane_sink(type, e, my_win, wb, src, sps_slot, pps_struct, zeros, out, &rec, &my_rows[lane], &dv, &dp);
                compact[k] = rec;
            } else {
                tr_n = parse_lane<kMode>(type, slice, e, structs + off, my_win, wb, src, sps_slot, pps_struct, zeros, out,
                                         trace ? trace + k * (uint64_t)trace_cap : nullptr, trace_cap, &my_rows[lane], nullptr, &dv, &dp);
                if (compact && slice) compact[k] = compact_of(*reinterpret_cast<const hevc_slice_header_t*>(structs + off));
            }
            if (slice) deps[k] = dp;
            if (dv) atomicOr(div_flag, 1u);        /* some slices are walked again, exactly (k4_fix) */
            parsed[k] = out;
            if (trace_count) trace_count[k] = tr_n;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* A batch of at most 64 NALs (the legacy single-NAL symbols parse one per call): plan, scan, clearing,
 * parameter sets, slices and the summary in ONE launch of one wavefront instead of ten launches. */
template <int kMode>
__global__ __launch_bounds__(64)
void k4_small(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n,
              ParsedNal* __restrict__ parsed, uint8_t* structs, uint64_t structs_cap,
              const uint8_t* __restrict__ zeros, const uint8_t* init_sps_slot, const uint8_t* init_pps,
              TraceRec* __restrict__ trace, uint32_t trace_cap, uint32_t* __restrict__ trace_count,
              RpsRow* __restrict__ own_rows /* 64 */, hbs_summary* __restrict__ sum, int sequential, uint32_t* __restrict__ div_flag)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[64 * kLaneWinStride];
    /* sequential (one NAL, the legacy symbols): the tables behind the SPS in force are THE tables, as in the reference */
    RpsTables* const seq_tables = (sequential && n == 1 && init_sps_slot)
        ? reinterpret_cast<RpsTables*>(const_cast<uint8_t*>(init_sps_slot) + round16(sizeof(hevc_sps_t))) : nullptr;
    const int lane = threadIdx.x;
    uint8_t* const my_win = win + (uint32_t)lane * kLaneWinStride;
    const bool have = (uint64_t)lane < n;

    /* plan (k4_plan) */
    hbs_nal_entry e;
    e.start = e.end = e.rbsp_off = 0; e.rbsp_len = 0; e.status = 0;
    ParsedNal p;
    p.rc = -1; p.nal_unit_type = -1; p.nal_layer_id = -1; p.nal_temporal_id_plus1 = -1;
    p.struct_off = ~0ull; p.slice_data_size = 0; p.slice_data_off = 0;
    uint64_t sz = 0;
    if (have) {
        e = idx[lane];
        if (!(e.status & HBS_ST_ERROR)) {
            nal_header_of(rbsp + e.rbsp_off, e.rbsp_len, p);
            sz = slot_bytes_of(p.nal_unit_type);
        }
    }
    const int type = p.nal_unit_type;
    /* slots and the parameter sets in force (k4_scan_*) */
    Scan3 x;
    x.sum = sz;
    x.sps = (have && type == HEVC_NAL_UNIT_TYPE_SPS_NUT) ? (long long)lane : -1;
    x.pps = (have && type == HEVC_NAL_UNIT_TYPE_PPS_NUT) ? (long long)lane : -1;
    Scan3 inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Scan3 t = scan3_shfl_up(inc, d);
        if (lane >= d) inc = scan3_join(t, inc);
    }
    Scan3 ex = scan3_shfl_up(inc, 1);
    if (lane == 0) { ex.sum = 0; ex.sps = -1; ex.pps = -1; }
    const unsigned long long total = __shfl(inc.sum, 63, 64);
    if (sz) p.struct_off = ex.sum;
    uint32_t err = 0;

    const bool slice = is_slice_type_nal(type);
    const bool pset = type == HEVC_NAL_UNIT_TYPE_VPS_NUT || type == HEVC_NAL_UNIT_TYPE_SPS_NUT || type == HEVC_NAL_UNIT_TYPE_PPS_NUT;
    bool active = structs != nullptr && have && type >= 0 && (slice || pset);
    if (active && p.struct_off + sz > structs_cap) { err = (uint32_t)(-HBS_E_CAPACITY); p.struct_off = ~0ull; active = false; }
    const uint64_t off = p.struct_off;
    uint32_t tr_n = 0;
    if (structs != nullptr) {
        /* the memset of every struct that gets parsed, and fresh rows for the slices' own short-term RPS */
        uint64_t todo = __ballot(active);
        while (todo) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint64_t oj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), j) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, j);
            zero_slot(structs + oj, slot_bytes_of(__builtin_amdgcn_readlane(type, j)), lane);
        }
        zero_slot(reinterpret_cast<uint8_t*>(own_rows), 64 * sizeof(RpsRow), lane);
        const uint8_t* src = rbsp + e.rbsp_off;
        const uint32_t wb = e.rbsp_len < kLaneWin ? e.rbsp_len : kLaneWin;
        if (active) stage_window(my_win, src, wb);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        /* where the parameter sets in force live: the lane that parsed them says */
        const uint64_t sps_off = (uint64_t)__shfl((unsigned long long)off, ex.sps >= 0 ? (int)ex.sps : 0, 64);
        const uint64_t pps_off = (uint64_t)__shfl((unsigned long long)off, ex.pps >= 0 ? (int)ex.pps : 0, 64);
        for (int pass = 0; pass < 2; ++pass) {
            if (active && (pass == 0) == pset) {
                const uint8_t* sps_slot = nullptr;
                const uint8_t* pps_struct = nullptr;
                if (slice) {
                    if (ex.sps >= 0) { if (sps_off != ~0ull) sps_slot = structs + sps_off; }
                    else sps_slot = init_sps_slot;
                    if (ex.pps >= 0) { if (pps_off != ~0ull) pps_struct = structs + pps_off; }
                    else pps_struct = init_pps;
                }
                int dv = 0;
                tr_n = parse_lane<kMode>(type, slice, e, structs + off, my_win, wb, src, sps_slot, pps_struct, zeros, p,
                                         trace ? trace + (uint64_t)lane * trace_cap : nullptr, trace_cap, &own_rows[lane], seq_tables, &dv);
                if (dv && n > 1) atomicOr(div_flag, 1u);
            }
            /* the slices read what the parameter-set lanes have just written */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (have) {
        parsed[lane] = p;
        if (trace_count) trace_count[lane] = tr_n;
    }
    const uint64_t any_err = __ballot(err != 0);
    if (lane == 0) {
        sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = 0; sum->stream_bytes = 0;
        sum->stop_reason = 0; sum->error = any_err ? HBS_E_CAPACITY : 0;
        sum->reserved[0] = total;                    /* struct arena bytes needed */
        sum->reserved[1] = sum->reserved[2] = 0;
    }
}

/* A whole batch the reference's way (hbs_ctx_set_sequential_parse): ONE wavefront walks the NALs in stream order,
 * lane 0 parsing, all lanes clearing structs; one set of derived RPS tables (hevc_stream.c:26-32) that every SPS and
 * every slice's own set writes into and reads from.  Orders of magnitude slower than k4_parse -- it exists so that
 * streams the spec forbids (a slice naming a set its SPS does not have) can be parsed exactly like the reference
 * when that matters.  Every SPS slot gets a snapshot of the tables as they stood right behind it. */
template <int kMode>
__global__ __launch_bounds__(64)
void k4_seq(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n,
            ParsedNal* __restrict__ parsed, uint8_t* structs, uint64_t structs_cap,
            const uint8_t* __restrict__ zeros, const uint8_t* init_sps_slot, const uint8_t* init_pps,
            TraceRec* __restrict__ trace, uint32_t trace_cap, uint32_t* __restrict__ trace_count,
            RpsTables* tables /* workspace */, hbs_summary* __restrict__ sum, const uint32_t* __restrict__ gate)
{
    /* behind the parallel parse: only when one of its slices saw that its answer depends on what NALs in front of
     * its SPS left in the tables (ParserT::diverged) */
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    /* one wavefront, so everything it hands from lane to lane stays on one compute unit: workgroup-scope fences (a wait
     * for the stores) are enough; with agent scope (write-back and invalidate of the caches, six times per NAL) the walk ran
     * at 92 k NAL/s */
    __shared__ __attribute__((aligned(16))) uint8_t win[64 * kLaneWinStride];
    const int lane = threadIdx.x;
    uint8_t* const my_win = win + (uint32_t)lane * kLaneWinStride;
    const uint64_t tbl_off = round16(sizeof(hevc_sps_t));
    /* the tables at "program start", or the ones behind the SPS the caller hands in */
    {
        uint4* d = reinterpret_cast<uint4*>(tables);
        const uint4* src = init_sps_slot ? reinterpret_cast<const uint4*>(init_sps_slot + tbl_off) : nullptr;
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(RpsTables) / 16); i += 64) d[i] = src ? src[i] : make_uint4(0, 0, 0, 0);
    }
    uint64_t off_run = 0, sps_off = ~0ull, pps_off = ~0ull;
    bool have_sps = false, have_pps = false;
    uint32_t err = 0;
    /* 64 NALs at a time.  What does not depend on the NALs in front -- entry, header, slot, the cleared struct, the first
     * bytes of the RBSP in LDS -- is done for the 64 at once, one NAL per lane (k4_small's plan); then the lanes take their
     * turn in stream order, each walking its own NAL against the one set of tables.  What remains is the walk itself: one
     * lane's dependent LDS and table reads, ~12 us per slice header (81 k NAL/s on the 4K30 batch of scripts/parse_time.py;
     * fetching entry, header and window per NAL when its turn came it was 77 k). */
#pragma unroll 1
    for (uint64_t k0 = 0; k0 < n; k0 += 64) {
        const uint64_t k = k0 + (uint64_t)lane;
        const bool have = k < n;
        hbs_nal_entry e;
        e.start = e.end = e.rbsp_off = 0; e.rbsp_len = 0; e.status = 0;
        ParsedNal p;
        p.rc = -1; p.nal_unit_type = -1; p.nal_layer_id = -1; p.nal_temporal_id_plus1 = -1;
        p.struct_off = ~0ull; p.slice_data_size = 0; p.slice_data_off = 0;
        uint64_t sz = 0;
        if (have) {
            e = idx[k];
            if (!(e.status & HBS_ST_ERROR)) {
                nal_header_of(rbsp + e.rbsp_off, e.rbsp_len, p);
                sz = slot_bytes_of(p.nal_unit_type);
            }
        }
        const int type = p.nal_unit_type;
        const bool slice = is_slice_type_nal(type);
        const bool pset = type == HEVC_NAL_UNIT_TYPE_VPS_NUT || type == HEVC_NAL_UNIT_TYPE_SPS_NUT || type == HEVC_NAL_UNIT_TYPE_PPS_NUT;
        unsigned long long inc = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        const uint64_t off = off_run + (inc - sz);
        if (sz) p.struct_off = off;
        bool active = have && type >= 0 && (slice || pset);
        if (active && off + sz > structs_cap) { err = (uint32_t)(-HBS_E_CAPACITY); p.struct_off = ~0ull; active = false; }
        off_run += (uint64_t)__shfl(inc, 63, 64);
        const uint64_t act_mask = __ballot(active);
        const uint64_t walk_mask = __ballot(have && type >= 0 && (slice || pset));      /* ... and those the capacity check turned away */
        for (uint64_t todo = act_mask; todo != 0; todo &= todo - 1) {
            const int j = (int)__builtin_ctzll(todo);
            const uint64_t oj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), j) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, j);
            zero_slot(structs + oj, slot_bytes_of(__builtin_amdgcn_readlane(type, j)), lane);
        }
        const uint8_t* const src = rbsp + e.rbsp_off;
        const uint32_t wb = e.rbsp_len < kLaneWin ? e.rbsp_len : kLaneWin;
        if (active) stage_window(my_win, src, wb);
        /* one wavefront, so everything it hands from lane to lane stays on one compute unit: workgroup-scope fences (a
         * wait for the stores) are enough; with agent scope (write-back and invalidate of the caches) the walk is slower */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t tr_n = 0;
        const int last = (n - k0 < 64) ? (int)(n - k0) : 64;
#pragma unroll 1
        for (int j = 0; j < last; ++j) {
            if (!((walk_mask >> j) & 1ull)) continue;                     /* wave-uniform */
            const bool active_j = ((act_mask >> j) & 1ull) != 0;
            const int type_j = __builtin_amdgcn_readlane(type, j);
            if (active_j && lane == j) {
                const uint8_t* sps_slot = nullptr;
                const uint8_t* pps_struct = nullptr;
                if (slice) {
                    if (have_sps) { if (sps_off != ~0ull) sps_slot = structs + sps_off; } else sps_slot = init_sps_slot;
                    if (have_pps) { if (pps_off != ~0ull) pps_struct = structs + pps_off; } else pps_struct = init_pps;
                }
                tr_n = parse_lane<kMode>(type, slice, e, structs + off, my_win, wb, src, sps_slot, pps_struct, zeros, p,
                                         trace ? trace + k * (uint64_t)trace_cap : nullptr, trace_cap, nullptr, tables);
            }
            if (type_j == HEVC_NAL_UNIT_TYPE_SPS_NUT || type_j == HEVC_NAL_UNIT_TYPE_PPS_NUT) {
                const uint64_t oj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), j) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, j);
                /* a set without room for its struct is still the set in force, with nothing to read from */
                if (type_j == HEVC_NAL_UNIT_TYPE_SPS_NUT) { have_sps = true; sps_off = active_j ? oj : ~0ull; }
                else { have_pps = true; pps_off = active_j ? oj : ~0ull; }
            }
            if (!active_j) continue;
            /* the next lane reads what this one left in the tables and the structs */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (type_j == HEVC_NAL_UNIT_TYPE_SPS_NUT) {          /* the tables as they stand now, next to the SPS */
                uint4* d = reinterpret_cast<uint4*>(structs + sps_off + tbl_off);
                const uint4* tsrc = reinterpret_cast<const uint4*>(tables);
                for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(RpsTables) / 16); i += 64) d[i] = tsrc[i];
            }
        }
        if (have) {
            parsed[k] = p;
            if (trace_count) trace_count[k] = tr_n;
        }
    }
    if (lane == 0) {
        sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = 0; sum->stream_bytes = 0;
        sum->stop_reason = 0; sum->error = err ? HBS_E_CAPACITY : 0;
        sum->reserved[0] = off_run;
        sum->reserved[1] = sum->reserved[2] = 0;
    }
}

/* ---- the exact re-walk of the slices that need it (hbs_parse_fix.h), gated on div_flag ------------------------------ */

__global__ __launch_bounds__(kFixBlock)
void k4_fix_masks(const ParsedNal* __restrict__ parsed, const uint8_t* __restrict__ structs, const uint32_t* __restrict__ deps, uint64_t n,
                  uint32_t* __restrict__ wmask, uint32_t* __restrict__ bsum, uint32_t* __restrict__ fix_count, const uint32_t* __restrict__ gate)
{
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    __shared__ uint32_t part[kFixBlock / 64];
    const uint64_t k = (uint64_t)blockIdx.x * kFixBlock + threadIdx.x;
    uint32_t m = 0u;
    if (k < n) { m = fix_wmask_of(parsed, structs, deps, k); wmask[k] = m; }
    uint32_t acc = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) acc |= __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = 0u;
        for (int w = 0; w < kFixBlock / 64; ++w) all |= part[w];
        bsum[blockIdx.x] = all;
        if (blockIdx.x == 0) { fix_count[0] = 0u; fix_count[1] = 0u; }
    }
}

__global__ __launch_bounds__(256)
void k4_fix_list(FixCtx c, uint32_t* __restrict__ list, uint32_t* __restrict__ fix_count, const uint32_t* __restrict__ gate)
{
    if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < c.n; k += (uint64_t)gridDim.x * blockDim.x)
        if (fix_is_affected(c, k)) list[atomicAdd(&fix_count[0], 1u)] = (uint32_t)k;
}

#ifndef HBS_FIX_ENABLED
#define HBS_FIX_ENABLED 1
#endif
constexpr unsigned kFixBlocks = 128;             /* x 64 lanes: slices walked again at the same time */
template <int kMode>
__global__ __launch_bounds__(64)
void k4_fix(FixCtx c, const uint32_t* __restrict__ list, uint32_t* __restrict__ fix_count, RpsRow* __restrict__ temps,
            TraceRec* __restrict__ trace, uint32_t trace_cap, uint32_t* __restrict__ trace_count, const uint32_t* __restrict__ gate,
            SliceCompact* __restrict__ compact = nullptr, uint8_t* __restrict__ tmp_structs = nullptr)
{
    if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const uint32_t count = fix_count[0];
    const uint32_t me = blockIdx.x * 64u + threadIdx.x;
    RpsRow* const my = temps + (uint64_t)me * kFixTemps;
    uint8_t* const my_struct = tmp_structs ? tmp_structs + (uint64_t)me * slot_bytes_of(HEVC_NAL_UNIT_TYPE_TRAIL_R) : nullptr;
    for (uint32_t q = me; q < count; q += gridDim.x * 64u) {
        const uint64_t k = list[q];
        uint32_t tr_n = 0;
#if HBS_FIX_ENABLED
        const bool ok = fix_slice<kMode>(c, k, my, trace ? trace + k * (uint64_t)trace_cap : nullptr, trace_cap, &tr_n, my_struct);
        if (ok && compact) {                                     /* the record again, from the struct the exact walk filled (its slot, or this lane's own) */
            const uint64_t so = c.parsed[k].struct_off;
            compact[k] = compact_of(*reinterpret_cast<const hevc_slice_header_t*>(so != ~0ull ? c.structs + so : my_struct));
        }
#else
        const bool ok = false; (void)my;
#endif
        if (!ok) atomicOr(&fix_count[1], 1u);                    /* a chain of slices deeper than kFixDepth: the whole batch in order (k4_seq) */
        else if (trace_count) trace_count[k] = tr_n;
    }
}

/* What the reference holds BEHIND the last NAL of the batch (hbs_parse_headers_state; the legacy symbols continue from it):
 * the SPS and PPS in force, and the 32 rows of the derived tables -- each row what its last writer left, found and evaluated
 * like the rows an out-of-spec slice reads (hbs_parse_fix.h), a lane per row.  One wavefront. */
__global__ __launch_bounds__(64)
void k4_state(FixCtx c, RpsRow* __restrict__ temps, uint8_t* __restrict__ scratch_sh /* 32 slice headers */,
              uint8_t* __restrict__ sps_slot_out, uint8_t* __restrict__ pps_out, hbs_summary* __restrict__ sum)
{
    const int lane = threadIdx.x;
    const uint64_t n = c.n;
    /* the parameter sets in force behind NAL n - 1 */
    long long cs = -1, cp = -1;
    if (n) {
        cs = c.ctx_sps[n - 1]; cp = c.ctx_pps[n - 1];
        const int t = c.parsed[n - 1].nal_unit_type;
        if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) cs = (long long)(n - 1);
        if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT) cp = (long long)(n - 1);
    }
    const uint64_t tbl_off = round16(sizeof(hevc_sps_t));
    const uint8_t* sps_src = cs >= 0 ? (c.parsed[cs].struct_off != ~0ull ? c.structs + c.parsed[cs].struct_off : c.zeros) : c.init_sps_slot;
    const uint8_t* pps_src = cp >= 0 ? (c.parsed[cp].struct_off != ~0ull ? c.structs + c.parsed[cp].struct_off : c.zeros) : c.init_pps;
    /* rows first (they may read the initial tables, which sps_slot_out may alias): into LDS-free temporaries, then out */
    RpsRow* const my = temps + (uint64_t)lane * (kFixDepth + 1);
    bool ok = true;
    if (lane < 32) {
        RowView v;
        ok = fix_resolve_row(c, lane, n, my, reinterpret_cast<hevc_slice_header_t*>(scratch_sh + (uint64_t)lane * round16(sizeof(hevc_slice_header_t))), v);
        RpsRow* keep = &my[kFixDepth];
        if (ok) {
            keep->NumDeltaPocs = v.nd; keep->NumNegativePics = v.nn; keep->NumPositivePics = v.np;
            for (int j = 0; j < 32; ++j) { keep->DeltaPocS0[j] = v.s0[j]; keep->UsedByCurrPicS0[j] = v.u0[j]; keep->DeltaPocS1[j] = v.s1[j]; keep->UsedByCurrPicS1[j] = v.u1[j]; }
        }
    }
    if (__ballot(!ok) != 0ull && lane == 0) sum->reserved[1] = 1;           /* a chain deeper than the re-walk follows: the rows it could not tell are left as they were */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    /* the structs (when they are not already where they go) */
    if (sps_src && sps_src != sps_slot_out)
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(hevc_sps_t) / 4); i += 64) reinterpret_cast<uint32_t*>(sps_slot_out)[i] = reinterpret_cast<const uint32_t*>(sps_src)[i];
    if (!sps_src)
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(hevc_sps_t) / 4); i += 64) reinterpret_cast<uint32_t*>(sps_slot_out)[i] = 0u;
    if (pps_src && pps_src != pps_out)
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(hevc_pps_t) / 4); i += 64) reinterpret_cast<uint32_t*>(pps_out)[i] = reinterpret_cast<const uint32_t*>(pps_src)[i];
    if (!pps_src)
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)(sizeof(hevc_pps_t) / 4); i += 64) reinterpret_cast<uint32_t*>(pps_out)[i] = 0u;
    if (lane < 32 && ok) {
        RpsTables* t = reinterpret_cast<RpsTables*>(sps_slot_out + tbl_off);
        const RpsRow* keep = &my[kFixDepth];
        t->NumDeltaPocs[lane] = keep->NumDeltaPocs; t->NumNegativePics[lane] = keep->NumNegativePics; t->NumPositivePics[lane] = keep->NumPositivePics;
        for (int j = 0; j < 32; ++j) {
            t->DeltaPocS0[lane][j] = keep->DeltaPocS0[j]; t->UsedByCurrPicS0[lane][j] = keep->UsedByCurrPicS0[j];
            t->DeltaPocS1[lane][j] = keep->DeltaPocS1[j]; t->UsedByCurrPicS1[lane][j] = keep->UsedByCurrPicS1[j];
        }
    }
}

/* ---- K5: syntax writers (write_hevc_nal_unit, hevc_stream.c:1249-1327, up to rbsp_to_nal) ---- */

__global__ void k5_slot_sizes(const ParsedNal* __restrict__ parsed, uint64_t n, unsigned long long* __restrict__ slot_size)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        slot_size[k] = (parsed[k].struct_off != ~0ull) ? slot_bytes_of(parsed[k].nal_unit_type) : 0ull;
}

/* one NAL per lane, as in k4_parse: lane l of a wavefront walks the syntax of NAL 64 c + l in write
 * mode into its own RBSP buffer (cleared beforehand in one go: the reference callocs it).
 * pass 0: parameter sets (an SPS re-derives the RPS tables behind it, as the reference's writer
 * refreshes its file-static ones) and the types that are not written at all; pass 1: slices. */
__global__ __launch_bounds__(256)
void k5_write(const ParsedNal* __restrict__ parsed, uint64_t n, int pass, uint8_t* __restrict__ structs,
              const long long* __restrict__ ctx_sps, const long long* __restrict__ ctx_pps,
              const uint8_t* __restrict__ zeros, const uint8_t* __restrict__ init_sps_slot, const uint8_t* __restrict__ init_pps,
              uint8_t* __restrict__ rbsp_out, uint32_t rbsp_cap, WrittenNal* __restrict__ written,
              RpsRow* __restrict__ own_rows /* 64 per wavefront of the grid */)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    RpsRow* const my_rows = own_rows + wave * 64;
    for (uint64_t chunk = wave; chunk * kParseLanes < n; chunk += nwaves) {
        const uint64_t k = chunk * kParseLanes + (uint64_t)lane;
        if (pass == 1) {                                       /* fresh rows for the slices' own short-term RPS, by the whole wave */
            uint4* q = reinterpret_cast<uint4*>(my_rows);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint32_t i = (uint32_t)lane; i < (uint32_t)(64 * sizeof(RpsRow) / 16); i += 64) q[i] = z;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (k >= n || lane >= kParseLanes) continue;
        const int type = parsed[k].nal_unit_type;
        const bool slice = is_slice_type_nal(type);
        const bool pset = type == HEVC_NAL_UNIT_TYPE_VPS_NUT || type == HEVC_NAL_UNIT_TYPE_SPS_NUT || type == HEVC_NAL_UNIT_TYPE_PPS_NUT;
        if ((slice || pset) && (pass == 0) != pset) continue;
        if (!(slice || pset) && pass != 0) continue;
        uint8_t* out = rbsp_out + k * (uint64_t)rbsp_cap;
        WrittenNal res;
        res.rc = -1; res.rbsp_size = 0; res.slice_data_size = 0; res.pad = 0;
        if ((slice || pset) && parsed[k].struct_off != ~0ull) {
            uint8_t* slot = structs + parsed[k].struct_off;
            ParserT<kModeWrite> ps;
            ps.b.win = out; ps.b.full = out; ps.b.win_bytes = 0; ps.b.size = rbsp_cap; ps.b.pos = 0;
            ps.b.tr = nullptr; ps.b.tr_cap = 0; ps.b.tr_n = 0; ps.b.wbuf = out;
            ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
            const hevc_sps_t* zero_sps = reinterpret_cast<const hevc_sps_t*>(zeros);
            const hevc_pps_t* zero_pps = reinterpret_cast<const hevc_pps_t*>(zeros);
            const hevc_sps_t* last_sps = zero_sps;
            const hevc_pps_t* last_pps = zero_pps;
            if (slice) {
                const long long cs = ctx_sps[k], cp = ctx_pps[k];
                if (cs >= 0 && parsed[cs].struct_off != ~0ull) {
                    last_sps = reinterpret_cast<const hevc_sps_t*>(structs + parsed[cs].struct_off);
                    ps.sps_rps = reinterpret_cast<const RpsTables*>(structs + parsed[cs].struct_off + round16(sizeof(hevc_sps_t)));
                } else if (cs < 0 && init_sps_slot) {
                    last_sps = reinterpret_cast<const hevc_sps_t*>(init_sps_slot);
                    ps.sps_rps = reinterpret_cast<const RpsTables*>(init_sps_slot + round16(sizeof(hevc_sps_t)));
                }
                if (cp >= 0 && parsed[cp].struct_off != ~0ull)
                    last_pps = reinterpret_cast<const hevc_pps_t*>(structs + parsed[cp].struct_off);
                else if (cp < 0 && init_pps)
                    last_pps = reinterpret_cast<const hevc_pps_t*>(init_pps);
                ps.own = &my_rows[lane];
            } else if (type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
                ps.out_rps = reinterpret_cast<RpsTables*>(slot + round16(sizeof(hevc_sps_t)));
            }
            write_one_nal(ps, type, parsed[k].nal_layer_id, parsed[k].nal_temporal_id_plus1, slot,
                          last_pps, last_sps, zero_pps, zero_sps, &res);
        }
        written[k] = res;
    }
}

/* The slots of the parameter sets, cleared by whole workgroups in front of the pass that parses them (round 5).  A hevc_vps_t is
 * 428 136 bytes and an SPS slot 93 KiB: cleared by the one wavefront that then parses the set (round 4), every VPS cost its
 * wavefront 418 rounds of stores before the first bit was read -- 67 us for the 627 parameter sets of BASELINE config 3, longer
 * than the 100 000 slices behind them took.  A workgroup looks at 256 consecutive NALs and clears what they need, 4 KiB a round. */
__global__ __launch_bounds__(256)
void k4_zero_psets(const ParsedNal* __restrict__ parsed, uint64_t n, uint8_t* __restrict__ structs, uint64_t structs_cap)
{
    __shared__ unsigned long long off[256];
    __shared__ uint32_t bytes[256];
    __shared__ uint32_t count;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k < n) {
        const int t = parsed[k].nal_unit_type;
        const uint64_t o = parsed[k].struct_off;
        if ((t == HEVC_NAL_UNIT_TYPE_VPS_NUT || t == HEVC_NAL_UNIT_TYPE_SPS_NUT || t == HEVC_NAL_UNIT_TYPE_PPS_NUT) && o != ~0ull &&
            o + slot_bytes_of(t) <= structs_cap) {
            const uint32_t i = atomicAdd(&count, 1u);
            off[i] = o; bytes[i] = (uint32_t)slot_bytes_of(t);
        }
    }
    __syncthreads();
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint32_t i = 0; i < count; ++i) {
        uint4* q = reinterpret_cast<uint4*>(structs + off[i]);
        for (uint32_t j = threadIdx.x; j < bytes[i] / 16u; j += 256u) q[j] = z;          /* slots are multiples of 16 */
    }
}

__global__ void k4_summary(uint64_t n, const unsigned long long* total, const uint32_t* err, hbs_summary* sum)
{
    sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = 0; sum->stream_bytes = 0;
    sum->stop_reason = 0; sum->error = -(int32_t)*err;
    sum->reserved[0] = *total;                       /* struct arena bytes needed */
    sum->reserved[1] = sum->reserved[2] = 0;
}

/* a compact parse has no sequential pass to fall back on: a chain of own sets deeper than the exact re-walk follows (never seen in
 * 12 000 fuzzed streams) is reported -- error HBS_E_DEPTH, reserved[1] = 1 -- and the caller takes hbs_parse_headers */
__global__ void k4_compact_verdict(const uint32_t* __restrict__ fix_count, hbs_summary* __restrict__ sum)
{
    if (fix_count[1]) { sum->reserved[1] = 1; if (sum->error == 0) sum->error = HBS_E_DEPTH; }
}

/* workgroups of the parse kernel: a wavefront per 64 NALs, at most kParseMaxBlocks workgroups */
unsigned parse_grid_blocks(uint64_t n)
{
    const uint64_t want = (n + 4 * kParseLanes - 1) / (4 * kParseLanes);
    return (unsigned)(want < 1 ? 1 : want > kParseMaxBlocks ? kParseMaxBlocks : want);
}
uint64_t parse_own_rows_bytes(uint64_t n) { return (uint64_t)parse_grid_blocks(n) * 4u * 64u * sizeof(RpsRow); }
uint64_t parse_fix_temps_bytes() { return (uint64_t)kFixBlocks * 64u * (uint64_t)kFixTemps * sizeof(RpsRow); }
uint64_t parse_fix_structs_bytes() { return (uint64_t)kFixBlocks * 64u * slot_bytes_of(HEVC_NAL_UNIT_TYPE_TRAIL_R); }

hipError_t launch_parse_headers(const ParseArgs& a, hipStream_t st)
{
    if (a.compact) {
        /* ---- compact parse (hbs_parse_headers_compact / hbs_parse_materialize): parameter sets into the struct arena as always,
         * slices walked into sinks (k4_parse<.., true>), except the listed ones; out-of-spec slices exactly, each by itself,
         * into a slot of the lane that walks it.  the counters (a.err, a.total, a.div_flag, a.fix_count) are cleared by the plan kernel: no memsets. */
        if (a.n == 0) {
            hipError_t e0 = hipMemsetAsync(a.total, 0, 512, st);           /* total, err, div_flag share 512 bytes of the workspace */
            if (e0 != hipSuccess) return e0;
            k4_summary<<<1, 1, 0, st>>>(a.n, a.total, a.err, a.summary);
            return hipGetLastError();
        }
        k4_plan<<<1024, 256, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.slot_size, a.deps, a.compact, a.total, a.err, a.div_flag, a.fix_count);
        if (a.want_n) {
            const uint64_t wb = (a.want_n + 255) / 256;
            k4_want<<<dim3((unsigned)(wb < 1024 ? wb : 1024)), 256, 0, st>>>(a.parsed, a.n, a.want_list, a.want_n, a.slot_size);
        }
        Scan3* part = reinterpret_cast<Scan3*>(a.scan_tmp);
        k4_scan_reduce<<<kScan4Blocks, 256, 0, st>>>(a.parsed, a.slot_size, a.n, part);
        k4_scan_parts<<<1, kScan4Blocks, 0, st>>>(part, a.total);
        k4_scan_apply<true><<<kScan4Blocks, 256, 0, st>>>(a.parsed, a.slot_size, a.n, part, a.ctx_sps, a.ctx_pps);
        if (!a.structs) { k4_summary<<<1, 1, 0, st>>>(a.n, a.total, a.err, a.summary); return hipGetLastError(); }     /* plan only */
        const unsigned pblocks = parse_grid_blocks(a.n);
        /* parameter sets, their slots cleared in front (and, when slices are wanted in full, the clearing of THEIR slots by the spare workgroups) */
        k4_zero_psets<<<dim3((unsigned)((a.n + 255) / 256)), 256, 0, st>>>(a.parsed, a.n, a.structs, a.structs_cap);
        k4_parse<kModeRead, false><<<pblocks + (a.want_n ? kZeroBlocks : 0u), 256, 0, st>>>(a.rbsp, a.index, a.n, 0, a.parsed, a.structs, a.structs_cap, a.ctx_sps, a.ctx_pps,
            a.zeros, a.initial_sps_slot, a.initial_pps, a.err, nullptr, 0, nullptr, a.own_rows, pblocks, a.div_flag, a.deps, a.compact, 1);
        k4_parse<kModeRead, true><<<pblocks, 256, 0, st>>>(a.rbsp, a.index, a.n, 1, a.parsed, a.structs, a.structs_cap, a.ctx_sps, a.ctx_pps,
            a.zeros, a.initial_sps_slot, a.initial_pps, a.err, nullptr, 0, nullptr, a.own_rows, pblocks, a.div_flag, a.deps, a.compact);
        if (a.want_n)
            k4_parse<kModeRead, false><<<pblocks, 256, 0, st>>>(a.rbsp, a.index, a.n, 1, a.parsed, a.structs, a.structs_cap, a.ctx_sps, a.ctx_pps,
                a.zeros, a.initial_sps_slot, a.initial_pps, a.err, nullptr, 0, nullptr, a.own_rows, pblocks, a.div_flag, a.deps, a.compact);
        k4_summary<<<1, 1, 0, st>>>(a.n, a.total, a.err, a.summary);
        FixCtx c;
        c.rbsp = a.rbsp; c.idx = a.index; c.n = a.n; c.parsed = a.parsed; c.structs = a.structs; c.structs_cap = a.structs_cap;
        c.ctx_sps = a.ctx_sps; c.ctx_pps = a.ctx_pps; c.zeros = a.zeros; c.init_sps_slot = a.initial_sps_slot; c.init_pps = a.initial_pps;
        c.deps = a.deps; c.wmask = a.wmask; c.bsum = a.bsum;
        const unsigned mblocks = (unsigned)((a.n + kFixBlock - 1) / kFixBlock);
        k4_fix_masks<<<mblocks, kFixBlock, 0, st>>>(a.parsed, a.structs, a.deps, a.n, a.wmask, a.bsum, a.fix_count, a.div_flag);
        k4_fix_list<<<mblocks < 1024u ? mblocks : 1024u, 256, 0, st>>>(c, a.fix_list, a.fix_count, a.div_flag);
        k4_fix<kModeRead><<<kFixBlocks, 64, 0, st>>>(c, a.fix_list, a.fix_count, a.fix_temps, nullptr, 0, nullptr, a.div_flag, a.compact, a.fix_structs);
        k4_compact_verdict<<<1, 1, 0, st>>>(a.fix_count, a.summary);
        return hipGetLastError();
    }
    if (a.sequential && a.structs && a.n > 1) {                           /* a whole batch, one NAL after the other */
        RpsTables* tables = reinterpret_cast<RpsTables*>(a.own_rows);
        static_assert(sizeof(RpsTables) <= 64 * sizeof(RpsRow), "the own-rows workspace of one wavefront holds the tables");
        if (a.trace)
            k4_seq<kModeTrace><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                 a.initial_pps, a.trace, a.trace_cap, a.trace_count, tables, a.summary, nullptr);
        else
            k4_seq<kModeRead><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                a.initial_pps, nullptr, 0, nullptr, tables, a.summary, nullptr);
        return hipGetLastError();
    }
    /* The parallel parse reads every slice against the tables of the SPS in front of it.  A slice whose answer depends on
     * more than that -- it names a set that SPS does not have, or rewrites one of its rows: streams the spec forbids --
     * raises div_flag.  Batches of up to 64 NALs are then walked again the reference's way, NAL after NAL with one set of
     * tables (k4_seq, gated on the flag: no host round trip; it returns at once on ordinary streams); larger ones walk
     * only the slices that need it (k4_fix, below). */
    hipError_t e = hipMemsetAsync(a.div_flag, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    RpsTables* const seq_tables_ws = reinterpret_cast<RpsTables*>(a.own_rows);
    auto exact_pass = [&]() {
        if (!a.structs || a.n < 2) return;
        if (a.trace)
            k4_seq<kModeTrace><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                 a.initial_pps, a.trace, a.trace_cap, a.trace_count, seq_tables_ws, a.summary, a.div_flag);
        else
            k4_seq<kModeRead><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                a.initial_pps, nullptr, 0, nullptr, seq_tables_ws, a.summary, a.div_flag);
    };
    if (a.n >= 1 && a.n <= 64 && !a.state_sps_slot_out) {
        if (a.trace)
            k4_small<kModeTrace><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                   a.initial_pps, a.trace, a.trace_cap, a.trace_count, a.own_rows, a.summary, a.sequential, a.div_flag);
        else
            k4_small<kModeRead><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                  a.initial_pps, nullptr, 0, nullptr, a.own_rows, a.summary, a.sequential, a.div_flag);
        exact_pass();
        return hipGetLastError();
    }
    e = hipMemsetAsync(a.err, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.total, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    if (a.n && a.trace_count) {
        e = hipMemsetAsync(a.trace_count, 0, a.n * sizeof(uint32_t), st);      /* NALs that are not parsed have no trace */
        if (e != hipSuccess) return e;
    }
    if (a.n) {
        k4_plan<<<1024, 256, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.slot_size, a.deps);
        Scan3* part = reinterpret_cast<Scan3*>(a.scan_tmp);
        k4_scan_reduce<<<kScan4Blocks, 256, 0, st>>>(a.parsed, a.slot_size, a.n, part);
        k4_scan_parts<<<1, kScan4Blocks, 0, st>>>(part, a.total);
        k4_scan_apply<true><<<kScan4Blocks, 256, 0, st>>>(a.parsed, a.slot_size, a.n, part, a.ctx_sps, a.ctx_pps);
        if (a.structs) {
            const unsigned pblocks = parse_grid_blocks(a.n);
            k4_zero_psets<<<dim3((unsigned)((a.n + 255) / 256)), 256, 0, st>>>(a.parsed, a.n, a.structs, a.structs_cap);
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned grid = pblocks + (pass == 0 ? kZeroBlocks : 0u);
                if (a.trace)
                    k4_parse<kModeTrace><<<grid, 256, 0, st>>>(a.rbsp, a.index, a.n, pass, a.parsed, a.structs, a.structs_cap, a.ctx_sps, a.ctx_pps, a.zeros, a.initial_sps_slot, a.initial_pps, a.err, a.trace, a.trace_cap, a.trace_count, a.own_rows, pblocks, a.div_flag, a.deps, nullptr, 1);
                else
                    k4_parse<kModeRead><<<grid, 256, 0, st>>>(a.rbsp, a.index, a.n, pass, a.parsed, a.structs, a.structs_cap, a.ctx_sps, a.ctx_pps, a.zeros, a.initial_sps_slot, a.initial_pps, a.err, nullptr, 0, nullptr, a.own_rows, pblocks, a.div_flag, a.deps, nullptr, 1);
            }
        }
    }
    k4_summary<<<1, 1, 0, st>>>(a.n, a.total, a.err, a.summary);
    if (a.structs && a.n) {
        /* A batch in which some slice raised div_flag: the slices whose rows have another last writer than their SPS are found
         * from the records every walk left (masks, list) and walked again, each by itself with the true rows handed in (k4_fix).
         * All three return at once on ordinary streams.  Only a chain of slices deeper than kFixDepth still sends the batch
         * through k4_seq (gated on fix_count[1]). */
        FixCtx c;
        c.rbsp = a.rbsp; c.idx = a.index; c.n = a.n; c.parsed = a.parsed; c.structs = a.structs; c.structs_cap = a.structs_cap;
        c.ctx_sps = a.ctx_sps; c.ctx_pps = a.ctx_pps; c.zeros = a.zeros; c.init_sps_slot = a.initial_sps_slot; c.init_pps = a.initial_pps;
        c.deps = a.deps; c.wmask = a.wmask; c.bsum = a.bsum;
        e = hipMemsetAsync(a.fix_count, 0, 2 * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
        const unsigned mblocks = (unsigned)((a.n + kFixBlock - 1) / kFixBlock);
        k4_fix_masks<<<mblocks, kFixBlock, 0, st>>>(a.parsed, a.structs, a.deps, a.n, a.wmask, a.bsum, a.fix_count, a.state_sps_slot_out ? nullptr : a.div_flag);
        const unsigned lblocks = mblocks < 1024u ? mblocks : 1024u;
        k4_fix_list<<<lblocks, 256, 0, st>>>(c, a.fix_list, a.fix_count, a.div_flag);
        if (a.trace)
            k4_fix<kModeTrace><<<kFixBlocks, 64, 0, st>>>(c, a.fix_list, a.fix_count, a.fix_temps, a.trace, a.trace_cap, a.trace_count, a.div_flag);
        else
            k4_fix<kModeRead><<<kFixBlocks, 64, 0, st>>>(c, a.fix_list, a.fix_count, a.fix_temps, nullptr, 0, nullptr, a.div_flag);
        if (a.n >= 2) {
            if (a.trace)
                k4_seq<kModeTrace><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                     a.initial_pps, a.trace, a.trace_cap, a.trace_count, seq_tables_ws, a.summary, a.fix_count + 1);
            else
                k4_seq<kModeRead><<<1, 64, 0, st>>>(a.rbsp, a.index, a.n, a.parsed, a.structs, a.structs_cap, a.zeros, a.initial_sps_slot,
                                                    a.initial_pps, nullptr, 0, nullptr, seq_tables_ws, a.summary, a.fix_count + 1);
        }
        if (a.state_sps_slot_out && a.state_pps_out) {
            /* (temps: 64 lanes x (kFixDepth + 1) rows at the front of the re-walk's; scratch headers behind them) */
            static_assert(64 * (kFixDepth + 1) * sizeof(RpsRow) + 32 * ((sizeof(hevc_slice_header_t) + 15) / 16 * 16) <= (size_t)kFixBlocks * 64 * kFixTemps * sizeof(RpsRow), "the state kernel fits the re-walk's temporaries");
            uint8_t* scratch = reinterpret_cast<uint8_t*>(a.fix_temps) + 64 * (kFixDepth + 1) * sizeof(RpsRow);
            k4_state<<<1, 64, 0, st>>>(c, a.fix_temps, scratch, a.state_sps_slot_out, a.state_pps_out, a.summary);
        }
    }
    return hipGetLastError();
}

hipError_t launch_write_headers(const WriteArgs& a, hipStream_t st)
{
    if (a.n) {
        Scan3* part = reinterpret_cast<Scan3*>(a.scan_tmp);
        k5_slot_sizes<<<1024, 256, 0, st>>>(a.parsed, a.n, a.slot_size);
        k4_scan_reduce<<<kScan4Blocks, 256, 0, st>>>(a.parsed, a.slot_size, a.n, part);
        k4_scan_parts<<<1, kScan4Blocks, 0, st>>>(part, a.total);
        k4_scan_apply<false><<<kScan4Blocks, 256, 0, st>>>(const_cast<ParsedNal*>(a.parsed), a.slot_size, a.n, part, a.ctx_sps, a.ctx_pps);
        hipError_t e = hipMemsetAsync(a.rbsp_out, 0, a.n * (uint64_t)a.rbsp_cap, st);
        if (e != hipSuccess) return e;
        for (int pass = 0; pass < 2; ++pass)
            k5_write<<<parse_grid_blocks(a.n), 256, 0, st>>>(a.parsed, a.n, pass, a.structs, a.ctx_sps, a.ctx_pps, a.zeros, a.initial_sps_slot,
                                                             a.initial_pps, a.rbsp_out, a.rbsp_cap, a.written, a.own_rows);
    }
    return hipGetLastError();
}

} // namespace hbs
