// itx_partition.hip — placeholder until the partition path lands (next commit).
#include "itx_partition.h"

struct ItxPartWork { int unused; };
int itx_part_create(const itx_table *, size_t, ItxPartWork **)
{
    itx_set_error("ITX_ACCUM_PARTITION is not built yet");
    return ITX_E_STATE;
}
void itx_part_destroy(ItxPartWork *) {}
int itx_part_run(ItxPartWork *, const ItxDevTable &, const ItxRunParams &, const ItxDevBatch &, size_t, int32_t *, uint64_t *,
                 uint32_t *, const ItxAccumLayout &, hipStream_t)
{
    return ITX_E_STATE;
}
