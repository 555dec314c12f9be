// mvs_step_plan -- prints the partition arithmetic of the C++ step host (mvs_step.hpp) as JSON, so that a CPU test can
// hold it against metagenome_vector_sketches_amd/parallel.py: block_plan / chunk_bounds / clip_blocks / shard rows.
//   mvs_step_plan <world> <block_rows> <n_total> <chunks> <first> <symmetric 0|1>
// No device is touched.
#include <cstdio>
#include <cstdlib>

#include "mvs_step.hpp"

int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: %s <world> <block_rows> <n_total> <chunks> <first> <symmetric>\n", argv[0]);
        return 1;
    }
    const int world = atoi(argv[1]);
    const long long block_rows = atoll(argv[2]), n_total = atoll(argv[3]);
    const int chunks = atoi(argv[4]);
    const double first = atof(argv[5]);
    const bool symmetric = atoi(argv[6]) != 0;
    const long long P = mvs_step::pad256(block_rows);
    printf("{\"P\": %lld, \"half_split\": %lld, \"chunks\": [", P, (long long)mvs_step::half_split(P));
    const auto cb = mvs_step::chunk_bounds(P, chunks, first);
    for (size_t k = 0; k < cb.size(); ++k) printf("%s[%lld, %lld]", k ? ", " : "", (long long)cb[k].first, (long long)cb[k].second);
    printf("], \"ranks\": [");
    for (int r = 0; r < world; ++r) {
        const auto rows = mvs_step::rank_rows(n_total, block_rows, r);
        const auto plan = mvs_step::block_plan(world, r, P, symmetric);
        printf("%s{\"rows\": [%lld, %lld], \"plan\": [", r ? ", " : "", (long long)rows.first, (long long)rows.second);
        for (size_t k = 0; k < plan.size(); ++k)
            printf("%s[%lld, %lld, %lld, %lld]", k ? ", " : "", (long long)plan[k].row_begin, (long long)plan[k].row_end,
                   (long long)plan[k].col_begin, (long long)plan[k].col_end);
        printf("], \"clipped\": [");
        for (size_t c = 0; c < cb.size(); ++c) {
            const auto cl = mvs_step::clip_blocks(plan, 1, P, cb[c].first, cb[c].second);
            printf("%s[", c ? ", " : "");
            for (size_t k = 0; k < cl.size(); ++k)
                printf("%s[%lld, %lld, %lld, %lld]", k ? ", " : "", (long long)cl[k].row_begin, (long long)cl[k].row_end,
                       (long long)cl[k].col_begin, (long long)cl[k].col_end);
            printf("]");
        }
        printf("]}");
    }
    printf("]}\n");
    return 0;
}
