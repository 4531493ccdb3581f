// gather_plan.hpp -- which transport operations one batch's gather consists of on one rank (every rank's record set to rank
// `dst`, one block per rank, ONE SET OF BLOCKS PER RECORD SET in rotation).  Pure host logic, no HIP: csrc/pipeline.hip executes the
// plan with ncclRecv / ncclSend / hipMemcpyAsync on the matching stream, tests/test_multigpu_cpu.py executes the SAME plan (through
// orbfe_pipeline_gather_plan) with torch.distributed send / recv between two gloo ranks on the CPU.
#pragma once
#include <stddef.h>

#include "../../include/orbfe.h"

namespace orbfe {

// ops[] in issue order; returns their number (at most world + 1), or -1 for an invalid argument / too small a capacity.
//   ORBFE_GATHER_RECV       receive rank `peer`'s record set into this rank's receive block at byte `offset` of the block buffer
//   ORBFE_GATHER_SEND       send this rank's record set `record_set` to rank `peer`
//   ORBFE_GATHER_COPY_OWN   (dst, world > 1) this rank's own record set into its block at `offset`, behind the group
// The block buffer of dst holds record_sets x world blocks of `nbytes`: batch b, written to record set s = b mod record_sets, lands in
// blocks [s * world, (s + 1) * world), so a consumer reads batch b while b + 1 .. b + record_sets - 1 arrive.
inline int gather_plan(int rank, int world, int dst, int record_set, int record_sets, size_t nbytes, orbfe_gather_op* ops, int capacity)
{
    if (!ops || world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || record_set < 0 || record_set >= record_sets) return -1;
    int n = 0;
    auto put = [&](int kind, int peer, size_t off) { if (n < capacity) ops[n] = orbfe_gather_op{kind, peer, (unsigned long long)off}; n++; };
    const size_t base = (size_t)record_set * (size_t)world * nbytes;
    if (rank == dst) {
        for (int r = 0; r < world; r++) {
            if (r == rank && world > 1) continue;      // its own block is a copy, not a message
            put(ORBFE_GATHER_RECV, r, base + (size_t)r * nbytes);
        }
        if (world == 1) put(ORBFE_GATHER_SEND, dst, 0); // the one-GPU box: the same kernels, to itself
        else put(ORBFE_GATHER_COPY_OWN, rank, base + (size_t)rank * nbytes);
    } else
        put(ORBFE_GATHER_SEND, dst, 0);
    return n <= capacity ? n : -1;
}

} // namespace orbfe
