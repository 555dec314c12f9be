// mvs_make_hashes -- writes a synthetic hash text file in the format `project_everything sketch` reads
// ("name: h1 h2 ...\n", SURVEY.md 8d: unique-ish u64 below 2^64 / 1000, clusters of 16 samples sharing 40 % of their
// hashes).  Test / measurement helper only: formatting half a billion numbers is far too slow in Python.
//   mvs_make_hashes <out.txt> <samples> <hashes per sample> [seed] [sorted]
// "sorted": every line's values ascending and unique (what this repository's `convert` writes); default: in generation
// order (what an unordered_set dump looks like to a reader: the reference's `convert`)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <out.txt> <samples> <hashes per sample> [seed]\n", argv[0]);
        return 1;
    }
    const long n = std::atol(argv[2]), h = std::atol(argv[3]);
    const uint64_t seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 1234;
    const bool sorted = argc > 5 && std::string(argv[5]) == "sorted";
    const uint64_t max_hash = 18446744073709552ULL;
    const long shared = (long)(0.4 * (double)h + 0.5);
    FILE* f = std::fopen(argv[1], "wb");
    if (!f) {
        std::perror(argv[1]);
        return 1;
    }
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 32 ? 32 : nt);
    const long batch = 4L * nt;                       // samples formatted per round, written in order
    std::vector<std::string> lines((size_t)batch);
    for (long s0 = 0; s0 < n; s0 += batch) {
        const long cnt = std::min(batch, n - s0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                char num[24];
                for (long k = t; k < cnt; k += nt) {
                    const long s = s0 + k;
                    std::string& line = lines[(size_t)k];
                    line.clear();
                    line.reserve((size_t)h * 19 + 32);
                    line += "sample" + std::to_string(s) + ":";
                    std::vector<uint64_t> vals((size_t)h);
                    for (long j = 0; j < h; ++j) {
                        // the first `shared` hashes come from the cluster's pool, the rest are private to the sample
                        const uint64_t key = j < shared ? ((uint64_t)(s / 16) << 40) ^ (uint64_t)j ^ 0xabcdef0000000000ULL
                                                        : ((uint64_t)s << 32) ^ (uint64_t)j;
                        vals[(size_t)j] = mix(mix(key) ^ seed) % max_hash;
                    }
                    if (sorted) {
                        std::sort(vals.begin(), vals.end());
                        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
                    }
                    for (uint64_t v : vals) {
                        int len = 0;
                        do {
                            num[len++] = (char)('0' + v % 10);
                            v /= 10;
                        } while (v);
                        line += ' ';
                        while (len) line += num[--len];
                    }
                    line += '\n';
                }
            });
        for (auto& th : pool) th.join();
        for (long k = 0; k < cnt; ++k)
            if (std::fwrite(lines[(size_t)k].data(), 1, lines[(size_t)k].size(), f) != lines[(size_t)k].size()) {
                std::perror("write");
                return 1;
            }
    }
    return std::fclose(f) == 0 ? 0 : 1;
}
