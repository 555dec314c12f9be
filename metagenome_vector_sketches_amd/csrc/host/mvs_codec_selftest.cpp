// mvs_codec_selftest -- CPU-only round-trip checks of mvs_codec.hpp and of the shard writer/reader in
// mvs_host.hpp (no device needed).  Exit code 0 = all good.
#include <random>
#include <set>
#include <sstream>

#include "mvs_host.hpp"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::cerr << "FAILED: " #cond " at line " << __LINE__ << std::endl;      \
            return 1;                                                                \
        }                                                                            \
    } while (0)

// `mvs_codec_selftest --parse <file> names|lines`: what read_hash_file() makes of a hash text file, one record per line as
// "<hex of the name>\t<count>\t<values>" -- the CPU suite compares it with what the reference binaries did with the same
// bytes (tests/golden/ref_parser.json).  MVS_HOST_NO_SIMD=1 takes the scalar parser.
static int dump_parse(const char* path, bool with_names) {
    mvs_host::HashSets sets;
    if (!mvs_host::read_hash_file(path, with_names, sets, 2)) {
        std::cerr << "cannot read " << path << std::endl;
        return 2;
    }
    static const char* hex = "0123456789abcdef";
    for (size_t i = 0; i + 1 < sets.offsets.size(); ++i) {
        std::string line;
        if (with_names)
            for (unsigned char ch : sets.names[i]) {
                line += hex[ch >> 4];
                line += hex[ch & 15];
            }
        line += '\t';
        line += std::to_string(sets.offsets[i + 1] - sets.offsets[i]);
        line += '\t';
        for (int64_t k = sets.offsets[i]; k < sets.offsets[i + 1]; ++k) {
            if (k > sets.offsets[i]) line += ' ';
            line += std::to_string(sets.hashes[(size_t)k]);
        }
        std::cout << line << "\n";
    }
    return 0;
}

int main(int argc, char* argv[]) {
    if (argc == 4 && std::string(argv[1]) == "--parse") return dump_parse(argv[2], std::string(argv[3]) == "names");
    std::mt19937_64 rng(12345);
    // compact_vector: widths 1..64, sizes incl. 0 and 1
    for (int width = 1; width <= 64; ++width) {
        for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(63), size_t(64), size_t(65), size_t(1000)}) {
            std::vector<uint64_t> v(n);
            const uint64_t mask = width == 64 ? ~0ULL : ((1ULL << width) - 1);
            for (auto& x : v) x = rng() & mask;
            if (n) v[n / 2] = mask;   // force the width
            mvs_codec::compact_vector cv;
            cv.build(v.begin(), v.size());
            std::stringstream ss;
            cv.save(ss);
            CHECK((uint64_t)ss.str().size() == cv.num_bytes());
            mvs_codec::compact_vector back;
            back.load(ss);
            CHECK(back.size() == n);
            for (size_t i = 0; i < n; ++i) CHECK(back[i] == v[i]);
        }
    }
    // rice_sequence: small deltas, large values, zeros, unsorted, single element
    for (int mode = 0; mode < 6; ++mode) {
        for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(64), size_t(65), size_t(129), size_t(5000)}) {
            std::vector<uint64_t> v(n);
            for (auto& x : v) {
                switch (mode) {
                    case 0: x = rng() % 4; break;
                    case 1: x = rng() % 1000; break;
                    case 2: x = 0; break;
                    case 3: x = rng() % (1ULL << 40); break;
                    case 4: x = (rng() % 100 == 0) ? (rng() % (1ULL << 30)) : (rng() % 3); break;
                    default: x = 1 + rng() % 100000; break;
                }
            }
            mvs_codec::rice_sequence rs;
            rs.encode(v.begin(), v.size());
            std::stringstream ss;
            rs.save(ss);
            CHECK((uint64_t)ss.str().size() == rs.num_bytes());
            mvs_codec::rice_sequence back;
            back.load(ss);
            CHECK(back.size() == n);
            std::vector<uint64_t> dec;
            back.decode(dec);
            CHECK(dec == v);
            for (size_t i = 0; i < n; i += (n > 100 ? 37 : 1)) CHECK(back.access(i) == v[i]);
            if (n) CHECK(back.access(n - 1) == v[n - 1]);
        }
    }
    // elias_fano: ascending sequences (dense, sparse, repeats, a single element, universe = last + 1 as the legacy
    // writer passes it); descending input is refused
    for (int mode = 0; mode < 4; ++mode) {
        for (size_t n : {size_t(1), size_t(2), size_t(64), size_t(65), size_t(129), size_t(5000)}) {
            std::vector<uint64_t> v(n);
            uint64_t cur = mode == 3 ? (1ULL << 33) : rng() % 50;
            for (auto& x : v) {
                x = cur;
                cur += mode == 0 ? 1 : (mode == 1 ? rng() % 100000 : (mode == 2 ? rng() % 2 : 1 + rng() % 7));
            }
            mvs_codec::elias_fano ef;
            ef.encode(v.begin(), v.size(), v.back() + 1);
            std::stringstream ss;
            ef.save(ss);
            CHECK((uint64_t)ss.str().size() == ef.num_bytes());
            mvs_codec::elias_fano back;
            back.load(ss);
            CHECK(back.size() == n && back.universe() == v.back() + 1);
            std::vector<uint64_t> dec;
            back.decode(dec);
            CHECK(dec == v);
            for (size_t i = 0; i < n; i += (n > 100 ? 37 : 1)) CHECK(back.access(i) == v[i]);
            CHECK(back.access(n - 1) == v[n - 1]);
        }
    }
    {
        std::vector<uint64_t> bad = {5, 3};
        mvs_codec::elias_fano ef;
        bool threw = false;
        try {
            ef.encode(bad.begin(), bad.size(), 6);
        } catch (const std::invalid_argument&) {
            threw = true;
        }
        CHECK(threw);
    }
    // shard writer/reader: rows with 1 entry, many entries, gaps; empty shard
    const std::string dir = argc > 1 ? std::string(argv[1]) : std::string("/tmp/mvs_codec_selftest/");
    std::filesystem::remove_all(dir);
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<mvs_cell> cells;
        if (variant > 0) {
            const int rows = variant == 1 ? 1 : 500;
            for (int r = 0; r < rows; ++r) {
                if (variant == 2 && r % 7 == 3) continue;   // absent rows
                const int cnt = variant == 1 ? 1 : (r % 5 == 0 ? 1 : 1 + (int)(rng() % 40));
                int col = (int)(rng() % 50);
                for (int k = 0; k < cnt; ++k) {
                    mvs_cell c{r * 3 + 10, col, 0, (int32_t)(13 + rng() % 243)};
                    cells.push_back(c);
                    col += 1 + (int)(rng() % (k % 3 == 0 ? 100000 : 5));
                }
            }
        }
        const std::string sub = dir + "shard_" + std::to_string(variant) + "/";
        mvs_host::write_shard(sub, cells.data(), cells.size());
        std::vector<mvs_cell> back;
        CHECK(mvs_host::read_shard(sub, back));
        CHECK(back.size() == cells.size());
        for (size_t i = 0; i < cells.size(); ++i)
            CHECK(back[i].row == cells[i].row && back[i].col == cells[i].col && back[i].q == cells[i].q);
    }
    {   // legacy int16 shard (elias_fano columns + round(dot / d) values, zstd where libzstd is present): round trip
        std::vector<mvs_cell> cells;
        const int d = 2048;
        for (int r = 0; r < 300; ++r) {
            if (r % 11 == 4) continue;
            const int cnt = r % 5 == 0 ? 1 : 1 + (int)(rng() % 30);
            int col = (int)(rng() % 50);
            for (int k = 0; k < cnt; ++k) {
                cells.push_back(mvs_cell{r * 2 + 1, col, (int32_t)(d * (10 + rng() % 4000) + rng() % d), 0});
                col += 1 + (int)(rng() % (k % 4 == 0 ? 50000 : 3));
            }
        }
        const std::string sub = dir + "legacy16/";
        CHECK(mvs_host::write_shard_legacy16(sub, cells.data(), cells.size(), d) > 0);
        if (mvs_host::Zstd::get().ok) {
            CHECK(std::filesystem::exists(sub + "matrix.bin.zst") && !std::filesystem::exists(sub + "matrix.bin"));
            CHECK(std::filesystem::exists(sub + "row_index.bin.zst") && !std::filesystem::exists(sub + "row_index.bin"));
        }
        std::vector<mvs_cell> back;
        CHECK(mvs_host::read_shard_legacy16(sub, back));
        CHECK(back.size() == cells.size());
        for (size_t i = 0; i < cells.size(); ++i)
            CHECK(back[i].row == cells[i].row && back[i].col == cells[i].col &&
                  back[i].dot == (int32_t)std::llround((double)cells[i].dot / d));
        std::vector<mvs_cell> none;
        CHECK(mvs_host::write_shard_legacy16(dir + "legacy16_empty/", none.data(), 0, d) == 0);
        back.clear();
        CHECK(mvs_host::read_shard_legacy16(dir + "legacy16_empty/", back) && back.empty());
    }
    {   // many rows: the multi-threaded writer produces the single-threaded writer's files byte for byte
        std::vector<mvs_cell> cells;
        for (int r = 0; r < 40000; ++r) {
            const int cnt = r % 5 == 0 ? 1 : 1 + (int)(rng() % 20);
            int col = (int)(rng() % 50);
            for (int k = 0; k < cnt; ++k) {
                cells.push_back(mvs_cell{r, col, 0, (int32_t)(13 + rng() % 243)});
                col += 1 + (int)(rng() % 1000);
            }
        }
        auto slurp = [](const std::string& f) {
            std::ifstream in(f, std::ios::binary);
            return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        };
        const mvs_host::ShardStats s1 = mvs_host::write_shard(dir + "one/", cells.data(), cells.size(), 1);
        const mvs_host::ShardStats s4 = mvs_host::write_shard(dir + "many/", cells.data(), cells.size(), 7);
        CHECK(s1.jac_space == s4.jac_space && s1.ngh_space == s4.ngh_space && s1.rows == s4.rows && s1.rows == 40000);
        for (const char* f : {"matrix.bin", "row_index.bin", "neighbor_start.bin"})
            CHECK(slurp(dir + "one/" + f) == slurp(dir + "many/" + f) && !slurp(dir + "one/" + f).empty());
        std::vector<mvs_cell> back;
        CHECK(mvs_host::read_shard(dir + "many/", back) && back.size() == cells.size());
        for (size_t i = 0; i < cells.size(); i += 97)
            CHECK(back[i].row == cells[i].row && back[i].col == cells[i].col && back[i].q == cells[i].q);
        // the streaming writer, fed the same cells as CSR pieces of uneven sizes (rows without cells in between, empty
        // pieces, 8-bit and 16-bit q arrays): the same three files byte for byte
        {
            mvs_host::ShardWriter w(dir + "stream/", 5);
            const int64_t n_rows_total = 40000 + 300;                    // trailing rows without cells
            std::vector<int64_t> first_cell((size_t)n_rows_total + 1, (int64_t)cells.size());
            for (size_t i = cells.size(); i-- > 0;) first_cell[(size_t)cells[i].row] = (int64_t)i;
            for (int64_t r = n_rows_total - 1; r >= 0; --r)
                if (first_cell[(size_t)r] == (int64_t)cells.size() || first_cell[(size_t)r] > first_cell[(size_t)r + 1])
                    first_cell[(size_t)r] = std::min(first_cell[(size_t)r], first_cell[(size_t)r + 1]);
            int piece = 0;
            for (int64_t r0 = 0; r0 < n_rows_total; ++piece) {
                const int64_t r1 = std::min<int64_t>(n_rows_total, r0 + (piece % 3 == 0 ? 1 : (piece % 3 == 1 ? 7777 : 0)));
                const int64_t c0 = first_cell[(size_t)r0], c1 = first_cell[(size_t)r1];
                std::vector<int64_t> rp((size_t)(r1 - r0) + 1);
                for (int64_t r = r0; r <= r1; ++r) rp[(size_t)(r - r0)] = first_cell[(size_t)r] - c0;
                std::vector<int32_t> col((size_t)(c1 - c0));
                std::vector<uint8_t> q8((size_t)(c1 - c0));
                std::vector<uint16_t> q16((size_t)(c1 - c0));
                for (int64_t k = c0; k < c1; ++k) {
                    col[(size_t)(k - c0)] = cells[(size_t)k].col;
                    q8[(size_t)(k - c0)] = (uint8_t)cells[(size_t)k].q;
                    q16[(size_t)(k - c0)] = (uint16_t)cells[(size_t)k].q;
                }
                mvs_row_block b{};
                b.row_begin = r0;
                b.row_end = r1;
                b.n_cells = c1 - c0;
                b.row_ptr = rp.data();
                b.col = col.data();
                if (piece % 2) b.q = q8.data();
                else b.q16 = q16.data();
                w.add(b);
                r0 = r1;
            }
            const mvs_host::ShardStats ss = w.finish();
            CHECK(ss.jac_space == s1.jac_space && ss.ngh_space == s1.ngh_space && ss.rows == s1.rows && w.cells() == cells.size());
            for (const char* f : {"matrix.bin", "row_index.bin", "neighbor_start.bin"})
                CHECK(slurp(dir + "one/" + f) == slurp(dir + "stream/" + f));
            // a writer that never reaches finish() (the comparison failed half way) leaves the shard that was there
            // untouched and nothing half-written behind
            {
                mvs_host::ShardWriter w2(dir + "stream/", 2);
                const int64_t rp2[2] = {0, 1};
                const int32_t col2[1] = {0};
                const uint8_t q2[1] = {255};
                mvs_row_block b2{};
                b2.row_begin = 0;
                b2.row_end = 1;
                b2.n_cells = 1;
                b2.row_ptr = rp2;
                b2.col = col2;
                b2.q = q2;
                w2.add(b2);
            }
            for (const char* f : {"matrix.bin", "row_index.bin", "neighbor_start.bin"}) {
                CHECK(slurp(dir + "one/" + f) == slurp(dir + "stream/" + f));
                CHECK(!std::filesystem::exists(dir + "stream/" + f + ".part"));
            }
        }
    }
    std::filesystem::remove_all(dir);
    // hash text parsing: dedup, stop at the first bad token, lines without ':' skipped
    {
        const std::string p = "/tmp/mvs_codec_selftest_hashes.txt";
        {
            std::ofstream f(p);
            f << "a: 3 1 2 3 3\nno colon here\nb:\nc: 18446744073709551615 7 18446744073709551616 9\nd: 5 x 6\ne: 12abc 4\r\n";
        }
        mvs_host::HashSets hs;
        CHECK(mvs_host::read_hash_file(p, true, hs, 2));
        CHECK(hs.names.size() == 5 && hs.names[0] == "a" && hs.names[1] == "b" && hs.names[4] == "e");
        CHECK(hs.offsets[1] - hs.offsets[0] == 3 && hs.hashes[0] == 1 && hs.hashes[2] == 3);
        CHECK(hs.offsets[2] - hs.offsets[1] == 0);
        CHECK(hs.offsets[3] - hs.offsets[2] == 2 && hs.hashes[(size_t)hs.offsets[3] - 1] == 18446744073709551615ULL);
        CHECK(hs.offsets[4] - hs.offsets[3] == 1 && hs.hashes[(size_t)hs.offsets[3]] == 5);
        CHECK(hs.offsets[5] - hs.offsets[4] == 1 && hs.hashes[(size_t)hs.offsets[4]] == 12);
        mvs_host::HashSets hl;
        CHECK(mvs_host::read_hash_file(p, false, hl, 1));
        CHECK(hl.offsets.size() == 7);   // six lines, every one a record
        // negative tokens: num_get<unsigned long> negates modulo 2^64 and carries on (-1 = 2^64 - 1, -0 = 0); a bare
        // or doubled sign ends the line
        {
            std::ofstream f(p);
            f << "n: 5 -1 7 -0\nm: 4 - 5\nk: +3 -+2 9\n";
        }
        mvs_host::HashSets hn;
        CHECK(mvs_host::read_hash_file(p, true, hn, 1));
        CHECK(hn.offsets[1] == 4 && hn.hashes[0] == 0 && hn.hashes[1] == 5 && hn.hashes[2] == 7 &&
              hn.hashes[3] == 18446744073709551615ULL);
        CHECK(hn.offsets[2] - hn.offsets[1] == 1 && hn.offsets[3] - hn.offsets[2] == 1 && hn.hashes[5] == 3);
        // binary CSR cache: round trip, then stale once the text changes
        CHECK(!mvs_host::load_csr_cache(p, hl));
        // what a killed run left behind goes away with the next cache write; a live writer's file (pid 1 stands in) stays
        const std::string dead = mvs_host::csr_cache_path(p) + ".part.2147483646", live = mvs_host::csr_cache_path(p) + ".part.1";
        for (const std::string& f : {dead, live}) std::ofstream(f) << "x";
        CHECK(mvs_host::write_csr_cache(p, hn));
        CHECK(!std::filesystem::exists(dead) && std::filesystem::exists(live));
        std::remove(live.c_str());
        mvs_host::HashSets hc;
        CHECK(mvs_host::load_csr_cache(p, hc));
        CHECK(hc.names == hn.names && hc.offsets == hn.offsets && hc.hashes.size() == hn.hashes.size());
        for (size_t i = 0; i < hn.hashes.size(); ++i) CHECK(hc.hashes[i] == hn.hashes[i]);
        {
            std::ofstream f(p, std::ios::app);
            f << "extra: 1\n";
        }
        mvs_host::HashSets hs2;
        CHECK(!mvs_host::load_csr_cache(p, hs2));                       // size / mtime no longer match
        {
            std::ofstream f(mvs_host::csr_cache_path(p), std::ios::binary | std::ios::trunc);
            f << "garbage";
        }
        CHECK(!mvs_host::load_csr_cache(p, hs2));
        std::remove(mvs_host::csr_cache_path(p).c_str());
        std::remove(p.c_str());
    }
    // long lines: strictly increasing text skips the sort, shuffled / duplicated text goes through the radix sort -- the
    // same sorted unique values either way (std::set as the referee); digit strings of 19, 20 and 21 digits around 2^64
    {
        std::mt19937_64 rng(11);
        for (int round = 0; round < 6; ++round) {
            const size_t n = round < 2 ? 300 : 20000;                     // below / above the radix sort's threshold
            std::vector<uint64_t> v(n);
            for (auto& x : v) x = round % 3 == 0 ? rng() : round % 3 == 1 ? rng() % 18446744073709552ULL : rng() % 5000;
            std::set<uint64_t> want(v.begin(), v.end());
            for (int order = 0; order < 3; ++order) {
                std::vector<uint64_t> t(v);
                if (order == 0) t.assign(want.begin(), want.end());      // increasing: the fast path
                if (order == 2) t.insert(t.end(), v.begin(), v.begin() + (long)(n / 10));   // duplicates
                std::string line;
                for (uint64_t x : t) line += " " + std::to_string(x);
                std::vector<uint64_t> got;
                mvs_host::parse_u64_tokens(line.data(), line.data() + line.size(), got);
                CHECK(got.size() == want.size() && std::equal(got.begin(), got.end(), want.begin()));
            }
        }
        const std::string edge = "9999999999999999999 18446744073709551615 09999999999999999999 18446744073709551616 5";
        std::vector<uint64_t> got;
        mvs_host::parse_u64_tokens(edge.data(), edge.data() + edge.size(), got);
        CHECK(got.size() == 2 && got[0] == 9999999999999999999ULL && got[1] == 18446744073709551615ULL);   // stops at 2^64
        const std::string lead = "000000000000000000000000000000007 00000000000000000000018446744073709551615 3";
        got.clear();
        mvs_host::parse_u64_tokens(lead.data(), lead.data() + lead.size(), got);
        CHECK(got.size() == 3 && got[0] == 3 && got[1] == 7 && got[2] == 18446744073709551615ULL);         // leading zeros
    }
    // the vectorised line parser against the scalar one, which defines the behaviour: random lines of digits and blanks
    // with every token length from 1 to 22 digits, values around 2^64, runs of blanks, line lengths around multiples of
    // 64 -- and lines with things the fast path must hand back (signs, tabs, letters, '\r', bytes above 127)
    {
        std::mt19937_64 rng(77);
        size_t fast_lines = 0;
        for (int round = 0; round < 4000; ++round) {
            std::string body;
            const int tokens = (int)(rng() % 40);
            const bool dirty = round % 5 == 4, wide = round % 3 == 0;       // wide: tokens the fast path hands back
            for (int t = 0; t < tokens; ++t) {
                body += std::string(1 + (rng() % 8 == 0 ? rng() % 3 : 0), ' ');
                const int kind = (int)(rng() % 12);
                if (kind == 0) body += "18446744073709551615";
                else if (kind == 1 && wide) body += rng() % 4 ? "18446744073709551616" : "18450000000000000000";
                else if (kind == 2) body += std::string(rng() % (wide ? 6 : 1), '0') + std::to_string(rng());
                else {
                    const int len = 1 + (int)(rng() % (wide ? 22 : 19));
                    for (int k = 0; k < len; ++k) body += (char)('0' + rng() % 10);
                }
                if (dirty && rng() % 6 == 0) body += std::string(1, "\t-+x\r\x80:"[rng() % 7]);
            }
            if (rng() % 3 == 0) body += " ";
            if (rng() % 7 == 0) body += "\r";
            while (rng() % 2 && body.size() % 64 != 0 && body.size() < 4096) body += " 7";     // some lengths on 64-byte marks
            const std::string padded = std::string(32, '#') + body + std::string(96, '\n');
            const char* b = padded.data() + 32;
            const char* e = b + body.size();
            const size_t room = mvs_host::count_token_starts(b, e);
            CHECK(room == mvs_host::count_token_starts_scalar(b, e));
            std::vector<uint64_t> want(room + 1, 0xdead), got(room + 1, 0xdead);
            bool inc_w = true, inc_g = true;
            const size_t nw = mvs_host::parse_u64_raw_scalar(b, e, want.data(), inc_w);
            const size_t ng = mvs_host::parse_u64_raw(b, e, padded.data(), padded.data() + padded.size(), got.data(), inc_g);
            CHECK(nw <= room && ng == nw && inc_w == inc_g);
            CHECK(std::equal(want.begin(), want.begin() + (long)nw, got.begin()));
#if defined(__x86_64__)
            if (mvs_host::host_has_avx2()) {
                size_t nf = 0;
                bool inc_f = true;
                std::vector<uint64_t> scratch(room + 1);
                const char* e2 = e;
                while (e2 > b && (e2[-1] == '\r' || e2[-1] == ' ')) --e2;
                fast_lines += mvs_host::parse_u64_line_avx2(b, e2, scratch.data(), nf, inc_f) ? 1 : 0;
            }
#endif
        }
#if defined(__x86_64__)
        if (mvs_host::host_has_avx2()) CHECK(fast_lines > 1500 && fast_lines < 3500);   // both outcomes are exercised
#endif
        // the sort's two routes: evenly spread values (top bits + insertion sweep) and values that crowd together (full sort)
        for (int shape = 0; shape < 4; ++shape) {
            std::vector<uint64_t> v(30000);
            for (auto& x : v)
                x = shape == 0 ? rng() : shape == 1 ? (rng() % 1000) << 40 | (rng() % 50) : shape == 2 ? (1ULL << 63) + rng() % 40000
                                                                                                       : rng() % 18446744073709552ULL;
            std::set<uint64_t> want(v.begin(), v.end());
            const size_t k = mvs_host::sort_unique_u64(v.data(), v.size());
            CHECK(k == want.size() && std::equal(v.begin(), v.begin() + (long)k, want.begin()));
        }
    }
    // read_hash_file writing the cache while it parses: the same sets as without, the cache loads; with lines that fall
    // short of their room (a duplicate, a bad token: the file is then written from the finished sets) and without (the
    // writers' own path); with a thread handed back and without; nothing is left behind when no cache is asked for
    for (int gaps = 0; gaps < 2; ++gaps) {
        const std::string p = "/tmp/mvs_codec_selftest_hashes2.txt";
        std::mt19937_64 rng(5);
        {
            std::ofstream f(p);
            for (int i = 0; i < 40; ++i) {
                f << "s" << i << ":";
                const int k = i == 7 ? 0 : 1 + (int)(rng() % 3000);
                for (int j = 0; j < k; ++j) f << " " << (i % 3 == 0 ? (uint64_t)j * 977 + 5 : (rng() % 18446744073709552ULL) | 1);
                if (gaps && i == 11) f << " 6 6 6";
                if (gaps && i == 13) f << " 12x 99 100";
                f << "\n";
            }
            f << "not a record\nlast: 3 2 1";
        }
        mvs_host::HashSets plain, direct, loaded;
        CHECK(mvs_host::read_hash_file(p, true, plain, 3));
        CHECK(!std::filesystem::exists(mvs_host::csr_cache_path(p)));
        if (gaps) {
            CHECK(mvs_host::read_hash_file(p, true, direct, 3, p));
        } else {
            std::thread writer;
            CHECK(mvs_host::read_hash_file(p, true, direct, 3, p, &writer));
            CHECK(writer.joinable());
            writer.join();
        }
        CHECK(!std::filesystem::exists(mvs_host::csr_cache_part_path(p)));
        CHECK(mvs_host::load_csr_cache(p, loaded));
        CHECK(plain.names.size() == 41 && plain.names[40] == "last" && plain.offsets[41] - plain.offsets[40] == 3);
        for (const mvs_host::HashSets* o : {&direct, &loaded}) {
            CHECK(o->names == plain.names && o->offsets == plain.offsets && o->hashes.size() == plain.hashes.size());
            for (size_t i = 0; i < plain.hashes.size(); ++i) CHECK(o->hashes[i] == plain.hashes[i]);
        }
        for (size_t i = 0; i + 1 < plain.offsets.size(); ++i)
            for (int64_t j = plain.offsets[i] + 1; j < plain.offsets[i + 1]; ++j) CHECK(plain.hashes[(size_t)j - 1] < plain.hashes[(size_t)j]);
        std::remove(mvs_host::csr_cache_path(p).c_str());
        std::remove(p.c_str());
    }
    // whole files: random records (names with and without ':', empty lines, "\r\n", trailing blanks, bad tokens, duplicates,
    // unordered and ordered values, long lines, a last line without newline) through read_hash_file with and without the
    // cache written behind the parsers, against the scalar line parser applied line by line
    {
        std::mt19937_64 rng(123);
        const std::string p = "/tmp/mvs_codec_selftest_hashes3.txt";
        for (int round = 0; round < 60; ++round) {
            std::string text;
            std::vector<std::string> want_names;
            std::vector<std::vector<uint64_t>> want_sets;
            const int lines = 1 + (int)(rng() % 40);
            for (int l = 0; l < lines; ++l) {
                const int kind = (int)(rng() % 10);
                std::string line;
                if (kind == 0) {
                    line = rng() % 2 ? "" : "a line without a colon 1 2 3";
                } else {
                    const std::string name = "s" + std::to_string(l) + (rng() % 5 == 0 ? " with blanks" : "");
                    line = name + ":";
                    const size_t n = kind == 1 ? 0 : kind == 2 ? 3000 + rng() % 3000 : rng() % 200;
                    uint64_t run = rng() % 1000;
                    for (size_t i = 0; i < n; ++i) {
                        uint64_t v = rng() % 3 == 0 ? (run += 1 + rng() % 100000) : rng() >> (rng() % 40);
                        if (round % 4 == 0) v = run += 1 + rng() % 1000;                     // an increasing file: the no-sort route
                        line += std::string(1 + (rng() % 20 == 0), ' ') + std::to_string(v);
                        if (rng() % 400 == 0) line += " " + std::to_string(v);                // a duplicate
                    }
                    if (rng() % 15 == 0) line += " 12x 5";
                    if (rng() % 15 == 0) line += " -7 8";
                    if (rng() % 10 == 0) line += "  ";
                    std::vector<uint64_t> vals(mvs_host::count_token_starts_scalar(line.data() + name.size() + 1, line.data() + line.size()) + 1);
                    bool inc = true;
                    const size_t k = mvs_host::parse_u64_raw_scalar(line.data() + name.size() + 1, line.data() + line.size(), vals.data(), inc);
                    vals.resize(k);
                    std::sort(vals.begin(), vals.end());
                    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
                    want_names.push_back(name);
                    want_sets.push_back(vals);
                }
                text += line;
                if (l + 1 < lines || rng() % 2) text += round % 7 == 3 ? "\r\n" : "\n";
            }
            {
                std::ofstream f(p, std::ios::binary);
                f << text;
            }
            for (int cached = 0; cached < 2; ++cached) {
                mvs_host::HashSets got;
                CHECK(mvs_host::read_hash_file(p, true, got, 1 + (unsigned)(rng() % 5), cached ? p : std::string()));
                mvs_host::HashSets from_cache;
                if (cached && !want_names.empty() && got.hashes.size()) CHECK(mvs_host::load_csr_cache(p, from_cache));
                for (const mvs_host::HashSets* o : {&got, &from_cache}) {
                    if (o == &from_cache && from_cache.names.empty()) continue;
                    CHECK(o->names == want_names && o->offsets.size() == want_names.size() + 1);
                    for (size_t i = 0; i < want_sets.size(); ++i) {
                        CHECK((size_t)(o->offsets[i + 1] - o->offsets[i]) == want_sets[i].size());
                        for (size_t j = 0; j < want_sets[i].size(); ++j) CHECK(o->hashes[(size_t)o->offsets[i] + j] == want_sets[i][j]);
                    }
                }
                std::remove(mvs_host::csr_cache_path(p).c_str());
            }
        }
        std::remove(p.c_str());
    }
    CHECK(mvs_host::format_g(56.46254) == "56.4625" && mvs_host::format_g(1234567.0) == "1.23457e+06");
    CHECK(mvs_host::format_g_float(-3.0f) == "-3" && mvs_host::format_g(0.0) == "0");
    std::cout << "mvs_codec_selftest ok" << std::endl;
    return 0;
}
