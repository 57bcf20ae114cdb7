// itx_partition.h — the ITX_ACCUM_PARTITION path: classify + key emit -> radix partition by consensus
// slot -> LDS histograms. See itx_partition.hip and DESIGN.md.
#pragma once
#include "itx_common.h"

struct ItxPartWork;
int itx_part_create(const itx_table *t, size_t batch_capacity, ItxPartWork **out);
void itx_part_destroy(ItxPartWork *w);
// Folds the per-stage HIP events recorded by itx_part_run since the last call into ms[5]
// (stream, count, plan, scatter, hist) and returns the keys emitted by the most recent batch.
void itx_part_fold_stats(ItxPartWork *w, double *ms, uint64_t *keys_last);
int itx_part_run(ItxPartWork *w, const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n,
                 int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, hipStream_t st);
