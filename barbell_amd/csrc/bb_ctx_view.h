// bb_ctx_view.h — what the satellite translation units (bb_trim.hip) may see of a bb_ctx.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "bb_common.h"

struct bb_trim_state;
struct bb_fastq_state;
struct bb_format_state;
struct bb_ctx_view {
    int device;
    hipStream_t stream;
    const bb_group_dev* d_groups;
    const uint32_t* d_label_ids;  // bb_filter_set's label id per histogram slot; null until it was called
    std::string* last_error;
    bb_trim_state** trim;
    bb_fastq_state** fastq;
    bb_format_state** format;
};
bb_ctx_view bb_ctx_get_view(bb_ctx* ctx);
void bb_trim_state_free(bb_trim_state* s);
void bb_fastq_state_free(bb_fastq_state* s);
void bb_format_state_free(bb_format_state* s);
