/* hbs_emit_launch.h -- host-visible launchers of K3 and the synthetic generator. */
#ifndef HBS_EMIT_LAUNCH_H
#define HBS_EMIT_LAUNCH_H

#include <hip/hip_runtime_api.h>
#include "hbs_common.h"

namespace hbs {

struct EmitArgs {
    const uint8_t* rbsp;              /* device RBSP arena                         */
    uint64_t rbsp_bytes;              /* informational (goes to the summary)       */
    const hbs_nal_entry* index_in;    /* rbsp_off/rbsp_len (+ start/end for gaps)  */
    uint64_t n;
    int gap_mode;                     /* 0: gaps from index_in; 1: synthetic rule  */
    uint8_t* out; uint64_t out_cap;
    hbs_nal_entry* index_out;         /* nullable                                  */
    hbs_summary* summary;
    /* workspace */
    unsigned long long* scan_tmp;     /* 1024 entries                              */
    unsigned long long* nal_total;    /* n                                         */
    unsigned long long* out_off;      /* n                                         */
    unsigned long long* total;        /* 1                                         */
    uint32_t* err;                    /* 1                                         */
    uint32_t* ticket;                 /* 1, on a cache line of its own             */
    unsigned long long* n_items;      /* 1: work items of the single-pass kernel (segments of <= 12 KiB)  */
    unsigned long long* items;        /* items_cap entries                                                */
    uint64_t items_cap;               /* emit_items_bound(n, payload bytes)                               */
    unsigned long long* desc;         /* items_cap / 12 + 1 look-back words                               */
    int grid_blocks;                  /* resident workgroups of the single-pass kernel (emit_grid_blocks) */
    int two_pass;                     /* 1: count / scan / emit as three steps; 0: the single pass; -1: picked on the device by density */
    unsigned long long* total_dense;  /* 1: the three-step path's total                                   */
    uint32_t* probe;                  /* 2: chunks sampled, chunks flagged                                */
    uint64_t clear_bytes;             /* desc .. probe are one stretch of the workspace this long: one clear per call */
    uint32_t* verdict_out;            /* 4 words owned by the context: the summary kernel leaves a copy of tflag[0..3] there (nullable) */
    uint32_t* tflag;                  /* 3, inside that stretch: the arena-tile kernel's eligibility (k3t_check) and its "gave up" */
    unsigned long long* first_k;      /* first_cap entries: per arena tile, the first NAL that begins in it               */
    uint64_t first_cap;
    int tiles;                        /* 0: never the arena-tile kernel; 1: when the index is eligible; 2: ... whatever the arena's size */
    int tile_blocks;                  /* resident workgroups of the arena-tile kernel (emit_tile_grid_blocks)          */
    /* dense tiles counted ahead of the tile kernel (hbs_emit.hip: k3t_sample) */
    uint32_t* cand_list;              /* cand_cap entries: tiles the sample lists                                       */
    uint32_t* cand_count;             /* 1, inside the cleared stretch                                                  */
    uint32_t* cand_ticket;            /* 1, inside the cleared stretch                                                  */
    uint64_t cand_cap;
    uint32_t* dz_table;               /* emit_dz_table_words() words per arena tile; an entry is valid when it carries call_no */
    uint32_t call_no;                 /* this call's number on its context (never 0)                                    */
    int first_static;                 /* hbs_ctx_set_device_exclusive: the tile kernel's first tile by workgroup number (else by ticket) */
};

struct SynthArgs {
    uint64_t seed; uint64_t n; int mode;
    uint8_t* rbsp; uint64_t rbsp_cap;
    hbs_nal_entry* index;
    hbs_summary* summary;
    unsigned long long* lens;         /* n */
    unsigned long long* offs;         /* n */
    unsigned long long* scan_tmp;     /* 1024 entries */
    unsigned long long* total;
    uint32_t* err;
};

int emit_grid_blocks(int device);
int emit_tile_grid_blocks(int device);
uint64_t emit_items_bound(uint64_t n, uint64_t payload_bytes);
uint64_t emit_desc_words(uint64_t items_cap);
uint64_t emit_dz_table_words();
hipError_t launch_emit_annexb(const EmitArgs& a, hipStream_t st);
/* the whole call in one launch of one workgroup (a few small NALs): no verdict words are written */
bool emit_takes_small_path(uint64_t n, uint64_t rbsp_bytes, int two_pass);
hipError_t launch_synth_rbsp(const SynthArgs& a, hipStream_t st);

} // namespace hbs
#endif
