// mvs_host.hpp -- host-side pieces shared by the drop-in executables: the reference's text/binary
// formats on both sides of the hot path.  Everything numeric is done by libmvs_hip.so (include/mvs_hip.h).
//
// Citations are relative to the reference root.
#ifndef MVS_HOST_HPP
#define MVS_HOST_HPP

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <iterator>
#include <memory>
#include <sstream>
#include <string>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include <cerrno>
#include <csignal>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../../include/mvs_hip.h"
#include "mvs_codec.hpp"

namespace mvs_host {

namespace fs = std::filesystem;

// ---------------------------------------------------------------------------------------------------
// hash text -> CSR
// ---------------------------------------------------------------------------------------------------
// Flat uint64 storage that is filled by the parser's workers (a std::vector would value-initialise
// gigabytes on one thread first).
struct HashBuffer {
    uint64_t* p = nullptr;
    size_t n = 0;
    void* map_base = nullptr;        // non-NULL: p points into a read-only file mapping of map_len bytes (the CSR cache)
    size_t map_len = 0;
    HashBuffer() = default;
    HashBuffer(const HashBuffer&) = delete;
    HashBuffer& operator=(const HashBuffer&) = delete;
    ~HashBuffer() { release(); }
    void release() {
        if (map_base) ::munmap(map_base, map_len);
        else free(p);
        p = nullptr;
        map_base = nullptr;
        n = map_len = 0;
    }
    bool reset(size_t count) {
        release();
        p = (uint64_t*)malloc(std::max<size_t>(8, count * sizeof(uint64_t)));
        n = p ? count : 0;
        return p != nullptr;
    }
    void adopt_mapping(void* base, size_t len, const uint64_t* first, size_t count) {
        release();
        map_base = base;
        map_len = len;
        p = const_cast<uint64_t*>(first);
        n = count;
    }
    const uint64_t* data() const { return p; }
    uint64_t* data() { return p; }
    size_t size() const { return n; }
    uint64_t operator[](size_t i) const { return p[i]; }
};

struct HashSets {
    std::vector<std::string> names;
    HashBuffer hashes;               // concatenated, unique within a sample (sorted)
    std::vector<int64_t> offsets;    // names.size() + 1
};

// ---------------------------------------------------------------------------------------------------
// Binary CSR cache of a hash text file: "<hash_file>.csr" next to it.  Parsing the text is what bounds
// `project_everything sketch` end to end (src/project_everything.cpp:264-281 parses serially; here all host threads
// parse, and it still is 10-100x the device time), so the parsed form -- names, offsets, unique sorted u64 values,
// exactly what read_hash_file() returns -- is kept and mapped on the next run (read_hash_file() writes it while it parses
// when asked to; write_csr_cache() writes one from sets in memory).  The cache is valid only for the
// text file it was made from (size and modification time are recorded); anything else falls back to the text.
//   header (64 B): magic "MVSCSR01", u64 text size, i64 text mtime (ns), u64 samples, u64 values, u64 name bytes
//   i64 offsets[samples + 1], u64 name_ends[samples], names, padding to 8, u64 values[]
// ---------------------------------------------------------------------------------------------------
struct CsrHeader {
    char magic[8];
    uint64_t text_size;
    int64_t text_mtime_ns;
    uint64_t samples, values, name_bytes;
    uint64_t reserved[2];
};
static_assert(sizeof(CsrHeader) == 64, "header layout");

inline std::string csr_cache_path(const std::string& hash_file) { return hash_file + ".csr"; }
// the file a cache is written to before it is renamed into place: one per writing process (two runs on the same hash file
// must not write into each other's file)
inline std::string csr_cache_part_path(const std::string& hash_file) {
    return csr_cache_path(hash_file) + ".part." + std::to_string((long)::getpid());
}

// ".part.<pid>" files next to `hash_file` whose writer no longer exists (a killed run leaves up to the size of the cache
// behind): removed before a new cache is written.  A pid that cannot be probed (another user's process) is left alone.
inline void remove_stale_cache_parts(const std::string& hash_file) {
    namespace fs = std::filesystem;
    std::error_code ec;
    const fs::path cache(csr_cache_path(hash_file));
    const fs::path dir = cache.parent_path().empty() ? fs::path(".") : cache.parent_path();
    const std::string prefix = cache.filename().string() + ".part.";
    for (fs::directory_iterator it(dir, ec), end; !ec && it != end; it.increment(ec)) {
        const std::string name = it->path().filename().string();
        if (name.size() <= prefix.size() || name.compare(0, prefix.size(), prefix) != 0) continue;
        const std::string tail = name.substr(prefix.size());
        if (tail.find_first_not_of("0123456789") != std::string::npos || tail.size() > 10) continue;
        const long pid = std::strtol(tail.c_str(), nullptr, 10);
        if (pid <= 0 || pid == (long)::getpid()) continue;
        if (::kill((pid_t)pid, 0) != 0 && errno == ESRCH) ::unlink(it->path().c_str());
    }
}

inline bool text_identity(const std::string& path, uint64_t& size, int64_t& mtime_ns) {
    struct stat st;
    if (::stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
    size = (uint64_t)st.st_size;
    mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    return true;
}

// Sort + unique of one sample's hashes when the text did not have them in strictly increasing order (the reference's
// `convert` writes an unordered_set's iteration order, src/project_everything.cpp:222-228).  The projection needs the
// set, not an order; sorted unique values are what the .csr cache is defined to hold, so the order is kept -- with an LSD
// radix sort on 11-bit digits over the bits that are actually set (FracMinHash values are < 2^64 / scaled: 5 passes),
// ~10 ns per value where std::sort takes 60 (that sort was 70 % of the first `sketch` run's parse stage).
// Returns the number of unique values, which are left sorted at v[0 .. return).
inline size_t sort_unique_u64(uint64_t* v, size_t n) {
    if (n < 256) {
        std::sort(v, v + n);
        return (size_t)(std::unique(v, v + n) - v);
    }
    uint64_t any = 0;
    for (size_t i = 0; i < n; ++i) any |= v[i];
    int bits = 0;
    while (bits < 64 && (any >> bits) != 0) ++bits;
    static thread_local std::vector<uint64_t> tmp;
    static thread_local std::vector<uint32_t> hist;
    tmp.resize(n);
    // LSD passes over the 11-bit digits at the given shifts (ascending); v holds the result
    auto radix = [&](const int* shifts, int passes) {
        hist.assign((size_t)passes * 2048, 0);
        for (size_t i = 0; i < n; ++i)
            for (int ps = 0; ps < passes; ++ps) ++hist[(size_t)ps * 2048 + ((v[i] >> shifts[ps]) & 2047)];
        uint64_t* src = v;
        uint64_t* dst = tmp.data();
        for (int ps = 0; ps < passes; ++ps) {
            uint32_t* h = hist.data() + (size_t)ps * 2048;
            uint32_t run = 0;
            bool one_bucket = false;
            for (int b = 0; b < 2048; ++b) {
                const uint32_t c = h[b];
                one_bucket = one_bucket || c == n;
                h[b] = run;
                run += c;
            }
            if (one_bucket) continue;                          // every value has the same digit here: nothing moves
            const int sh = shifts[ps];
            for (size_t i = 0; i < n; ++i) dst[h[(src[i] >> sh) & 2047]++] = src[i];
            std::swap(src, dst);
        }
        if (src != v) memcpy(v, src, n * sizeof(uint64_t));
    };
    bool sorted = false;
    if (bits > 22 && n < ((size_t)1 << 21)) {
        // Hash values are spread evenly: the top 22 bits alone put fewer than 2^21 of them in order but for a few
        // neighbours, which one insertion sweep settles -- two passes instead of five or six.  A sweep that has to move
        // more than a few places per value (values that crowd together) is given up for the full sort.
        const int top[2] = {bits - 22, bits - 11};
        radix(top, 2);
        size_t moved = 0;
        const size_t budget = 4 * n;
        size_t i = 1;
        for (; i < n && moved <= budget; ++i) {
            const uint64_t x = v[i];
            size_t j = i;
            while (j > 0 && v[j - 1] > x) {
                v[j] = v[j - 1];
                --j;
            }
            v[j] = x;
            moved += i - j;
        }
        sorted = i == n && moved <= budget;
    }
    if (!sorted) {
        int all[6];
        const int passes = std::max(1, (bits + 10) / 11);
        for (int ps = 0; ps < passes; ++ps) all[ps] = 11 * ps;
        radix(all, passes);
    }
    return (size_t)(std::unique(v, v + n) - v);
}

// An upper bound of the number of values parse_u64_raw() extracts from [p, end): every extracted value consumes one whole
// run of decimal digits (an optional sign in front of it), so there are at most as many values as digit runs -- "12-3" is
// two values to `iss >> hash` (12, then -3 modulo 2^64), "1+2+3" three.  Exact for well-formed lines without duplicates.
inline size_t count_token_starts_scalar(const char* p, const char* end) {
    size_t count = 0;
    bool prev_digit = false;
    for (; p < end; ++p) {
        const bool digit = (unsigned)(*p - '0') <= 9u;
        count += (size_t)(digit && !prev_digit);
        prev_digit = digit;
    }
    return count;
}

// Parse whitespace separated unsigned 64-bit integers the way `while (iss >> hash)` does
// (src/project_everything.cpp:275-279, src/standalone_projection.cpp:32-35): stop at the first token
// that is not a number in range.  Values are stored at dst in text order (the caller provides room for
// count_token_starts() of them); `increasing` is cleared when they are not strictly increasing.
inline size_t parse_u64_raw_scalar(const char* p, const char* end, uint64_t* dst, bool& increasing) {
    size_t n = 0;
    uint64_t prev = 0;
    auto push = [&](uint64_t v) {
        increasing = increasing && (n == 0 || v > prev);
        prev = v;
        dst[n++] = v;
    };
    while (true) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
        if (p >= end) break;
        // one optional sign; num_get<unsigned long> accepts '-' and negates modulo 2^64 ("-1" reads 2^64 - 1)
        const bool negate = *p == '-';
        if (*p == '+' || *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') break;
        // up to 19 digits cannot overflow (10^19 < 2^64): only a 20th digit needs the check
        uint64_t v = 0;
        bool overflow = false;
        const char* q = p;
        const char* lim = end - p > 19 ? p + 19 : end;
        while (q < lim && (unsigned)(*q - '0') <= 9u) v = v * 10 + (uint64_t)(*q++ - '0');
        while (q < end && (unsigned)(*q - '0') <= 9u) {
            const uint64_t dgt = (uint64_t)(*q - '0');
            if (v > (UINT64_MAX - dgt) / 10) overflow = true;
            v = v * 10 + dgt;
            ++q;
        }
        p = q;
        if (overflow) break;                                   // failbit in the reference
        if (negate) v = 0 - v;
        // the extraction stops at the first character that is not a digit and the NEXT one starts right there: "12abc"
        // gives 12 and then fails at 'a', "12-3" gives 12 and then -3 (modulo 2^64), "1+2" gives 1 and 2
        // (tests/golden/ref_parser.json: the reference binaries on such lines)
        push(v);
    }
    return n;
}

#if defined(__x86_64__)
#define MVS_HOST_AVX2 __attribute__((target("avx2,bmi,popcnt")))
MVS_HOST_AVX2 inline size_t count_token_starts_avx2(const char* p, const char* end) {
    size_t count = 0;
    uint32_t carry = 0;                                       // was the byte in front a digit?
    const __m256i below0 = _mm256_set1_epi8('0' - 1), above9 = _mm256_set1_epi8('9' + 1);
    for (; end - p >= 32; p += 32) {
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
        const uint32_t w = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi8(c, below0), _mm256_cmpgt_epi8(above9, c)));
        count += (size_t)__builtin_popcount(w & ~((w << 1) | carry));
        carry = w >> 31;
    }
    bool prev_digit = carry != 0;
    for (; p < end; ++p) {
        const bool digit = (unsigned)(*p - '0') <= 9u;
        count += (size_t)(digit && !prev_digit);
        prev_digit = digit;
    }
    return count;
}

// the value of the (up to 16) digits that end at pos: three multiply-adds
MVS_HOST_AVX2 inline uint64_t last_digits16_avx2(const char* pos, size_t len) {
    alignas(16) static const uint8_t keep_tail[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                      255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255};
    __m128i c = _mm_sub_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(pos - 16)), _mm_set1_epi8('0'));
    c = _mm_and_si128(c, _mm_loadu_si128(reinterpret_cast<const __m128i*>(keep_tail + len)));
    const __m128i t1 = _mm_maddubs_epi16(c, _mm_set1_epi16(0x010a));          // bytes (10, 1): pairs of digits -> 0..99
    const __m128i t2 = _mm_madd_epi16(t1, _mm_set1_epi32(0x00010064));        // words (100, 1): -> 0..9999
    const __m128i t3 = _mm_packus_epi32(t2, t2);
    const __m128i t4 = _mm_madd_epi16(t3, _mm_set1_epi32(0x00012710));        // words (10000, 1): -> 0..99999999
    const uint64_t both = (uint64_t)_mm_cvtsi128_si64(t4);
    return (both & 0xffffffffu) * 100000000ull + (both >> 32);
}

// digits [tok, pos) -> value; false: more than 20 digits or a value >= 2^64
MVS_HOST_AVX2 inline bool digits_to_u64_avx2(const char* tok, const char* pos, uint64_t& v) {
    const size_t len = (size_t)(pos - tok);
    if (len <= 16) {
        v = last_digits16_avx2(pos, len);
        return true;
    }
    if (len > 20) return false;
    uint64_t hi = 0;
    for (const char* q = tok; q < pos - 16; ++q) hi = hi * 10 + (uint64_t)(*q - '0');
    const uint64_t lo = last_digits16_avx2(pos, 16);
    if (hi > 1844 || (hi == 1844 && lo > 6744073709551615ull)) return false;
    v = hi * 10000000000000000ull + lo;
    return true;
}

// The common line -- decimal digits and blanks, nothing else, tokens of at most 20 digits in range -- without a
// branch per character: 64 bytes give one bit mask of the non-digits, every set bit ends a token, and a token's last 16
// digits are converted with three multiply-adds (the wider ones, up to four leading digits, by hand).  Returns false
// when the line holds anything else (a sign, a tab, a letter, 21 digits, a value >= 2^64): the caller then parses the
// whole line with parse_u64_raw_scalar(), which defines the behaviour.  The loads reach up to 63 bytes past `end` and 16
// bytes in front of a token: the caller guarantees map_begin + 16 <= p and end + 64 <= map_end.
MVS_HOST_AVX2 inline bool parse_u64_line_avx2(const char* p, const char* end, uint64_t* dst, size_t& n_out, bool& increasing) {
    const __m256i below0 = _mm256_set1_epi8('0' - 1), above9 = _mm256_set1_epi8('9' + 1), blank = _mm256_set1_epi8(' ');
    size_t n = 0;
    uint64_t prev = 0, v = 0;
    bool inc = true;
    const char* tok = p;
    for (const char* b = p; b < end; b += 64) {
        const __m256i c0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b));
        const __m256i c1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + 32));
        const uint64_t d0 = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi8(c0, below0), _mm256_cmpgt_epi8(above9, c0)));
        const uint64_t d1 = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi8(c1, below0), _mm256_cmpgt_epi8(above9, c1)));
        const uint64_t s0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(c0, blank));
        const uint64_t s1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(c1, blank));
        const size_t len = (size_t)(end - b);
        const uint64_t valid = len >= 64 ? ~0ull : ((1ull << len) - 1);
        uint64_t nd = ~(d0 | (d1 << 32)) & valid;
        if (nd != ((s0 | (s1 << 32)) & valid)) return false;
        if (len < 64) nd |= 1ull << len;                       // the end of the line ends the last token
        while (nd) {
            const char* pos = b + __builtin_ctzll(nd);
            nd &= nd - 1;
            if (pos > tok) {
                if (!digits_to_u64_avx2(tok, pos, v)) return false;
                inc = inc && (n == 0 || v > prev);
                prev = v;
                dst[n++] = v;
            }
            tok = pos + 1;
        }
    }
    if (tok < end) {                                           // a line whose length is a multiple of 64
        if (!digits_to_u64_avx2(tok, end, v)) return false;
        inc = inc && (n == 0 || v > prev);
        dst[n++] = v;
    }
    n_out = n;
    increasing = increasing && inc;
    return true;
}

inline bool host_has_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt") &&
                            !getenv("MVS_HOST_NO_SIMD");
    return yes;
}
#else
inline bool host_has_avx2() { return false; }
#endif

inline size_t count_token_starts(const char* p, const char* end) {
#if defined(__x86_64__)
    if (host_has_avx2()) return count_token_starts_avx2(p, end);
#endif
    return count_token_starts_scalar(p, end);
}

// [p, end) -> dst (room for count_token_starts(p, end) values), in text order.  map_begin / map_end: the readable range
// around the line (the vectorised path reads 16 bytes in front of a token and up to 63 bytes past the line).
inline size_t parse_u64_raw(const char* p, const char* end, const char* map_begin, const char* map_end, uint64_t* dst, bool& increasing) {
#if defined(__x86_64__)
    if (host_has_avx2() && p - map_begin >= 16 && map_end - end >= 64) {
        const char* e = end;
        while (e > p && (e[-1] == '\r' || e[-1] == ' ')) --e;             // "\r\n" files; trailing blanks
        size_t n = 0;
        bool inc = increasing;
        if (parse_u64_line_avx2(p, e, dst, n, inc)) {
            increasing = inc;
            return n;
        }
    }
#endif
    return parse_u64_raw_scalar(p, end, dst, increasing);
}

// The sorted unique values of a line, APPENDED to `out` (a line whose values are strictly increasing -- what this
// repository's `convert` writes -- is unique and sorted as it stands and skips the sort).
inline void parse_u64_tokens(const char* p, const char* end, std::vector<uint64_t>& out) {
    const size_t first = out.size();
    out.resize(first + count_token_starts(p, end));
    bool increasing = true;
    size_t n = parse_u64_raw(p, end, p, end, out.data() + first, increasing);     // no slack around a bare string: scalar
    if (!increasing) n = sort_unique_u64(out.data() + first, n);
    out.resize(first + n);
}

inline bool write_csr_cache(const std::string& hash_file, const HashSets& sets, unsigned threads = 0);

// One record per line.  with_names: "name: h1 h2 ..." (lines without ':' are skipped,
// src/project_everything.cpp:267-270); otherwise every line is a hash list
// (src/standalone_projection.cpp:28-35).
// cache_for: when not empty, the binary cache "<cache_for>.csr" (above) is written WHILE the text is parsed: a line's
// values have their final place in the flat array -- and in the file -- before the line is parsed (pass 1 counts), so a
// few writer threads copy finished runs of lines into the page cache behind the parsers (4 GB for 10k samples of 50k
// hashes, at the ~5 GB/s the kernel takes them: started after the parse, beside the projection, that copy was what a first
// `sketch` run waited for at its end).  If a line turns out shorter than its room (a duplicate, a bad token) the file is
// written from the finished sets instead.  cache_writer: the thread that completes the cache (it may outlive this call
// and reads `out`: join it before `out` goes away); without one the call returns when the cache is complete.
// (Parsing straight into a MAP_SHARED mapping of the cache file -- no copy at all -- took 8 s instead of 0.5 s on the GPU
// box's file system: a write fault per 4 KiB page of a file mapping, 16 threads deep.)
inline bool read_hash_file(const std::string& path, bool with_names, HashSets& out, unsigned threads = 0,
                           const std::string& cache_for = std::string(), std::thread* cache_writer = nullptr) {
    // the file is mapped, not copied: hash lists run to gigabytes of text
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return false;
    }
    const size_t size = (size_t)st.st_size;
    const char* buf = nullptr;
    if (size) {
        void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            return false;
        }
        ::madvise(m, size, MADV_SEQUENTIAL);
        buf = (const char*)m;
    }
    ::close(fd);
    // the mapping is readable up to the end of its last page (zero filled past the end of the file)
    const size_t page = (size_t)std::max<long>(4096, sysconf(_SC_PAGESIZE));
    const char* const map_end = buf ? buf + (size + page - 1) / page * page : nullptr;
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    const bool lap_on = getenv("MVS_STAGE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t = std::chrono::steady_clock::now();
        if (lap_on) std::cerr << "[stage]   read_hash_file: " << what << " " << std::chrono::duration<double>(t - lap_t).count() << " s" << std::endl;
        lap_t = t;
    };
    struct Unmap {
        const char* b;
        size_t n;
        ~Unmap() { if (b && n) ::munmap((void*)b, n); }
    } unmap_text{buf, size};

    // line table: each worker scans one slice of the file for '\n'.  std::getline yields a final record
    // without '\n' if there is text after the last newline and no extra record for a trailing newline.
    struct Rec { size_t b, e, colon; };
    const unsigned tl = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, size >> 22));
    std::vector<std::vector<size_t>> nl(tl);
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < tl; ++t)
            pool.emplace_back([&, t]() {
                size_t pos = size * t / tl;
                const size_t end = size * (t + 1) / tl;
                while (pos < end) {
                    const char* q = (const char*)memchr(buf + pos, '\n', end - pos);
                    if (!q) break;
                    nl[t].push_back((size_t)(q - buf));
                    pos = (size_t)(q - buf) + 1;
                }
            });
        for (auto& th : pool) th.join();
    }
    std::vector<Rec> recs;
    size_t pos = 0;
    auto add_line = [&](size_t b, size_t e) {
        if (with_names) {
            const char* c = (const char*)memchr(buf + b, ':', e - b);      // no ':' -> skipped (:267-270)
            if (c) recs.push_back({b, e, (size_t)(c - buf)});
        } else {
            recs.push_back({b, e, b - 1});
        }
    };
    for (unsigned t = 0; t < tl; ++t)
        for (size_t e : nl[t]) {
            add_line(pos, e);
            pos = e + 1;
        }
    if (pos < size) add_line(pos, size);
    lap("line table");

    const size_t n = recs.size();
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n));
    // an exception in a worker must not end in std::terminate: it is carried to the caller
    std::exception_ptr worker_error;
    std::mutex worker_mutex;
    auto run = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                try {
                    body(t);
                } catch (...) {
                    std::lock_guard<std::mutex> lock(worker_mutex);
                    if (!worker_error) worker_error = std::current_exception();
                }
            });
        for (auto& th : pool) th.join();
    };
    // pass 1: an upper bound of every line's value count (exact for well-formed lines without duplicates) fixes where the
    // line's values go in the flat array, so that pass 2 can parse them in place (no buffer per sample, no copy into the
    // flat array afterwards)
    std::vector<int64_t> room(n + 1, 0);
    std::atomic<size_t> next_line{0};
    const size_t grain = std::max<size_t>(1, n / ((size_t)threads * 32));
    auto for_lines = [&](auto&& body) {
        next_line = 0;
        run([&](unsigned) {
            for (size_t i0 = next_line.fetch_add(grain); i0 < n; i0 = next_line.fetch_add(grain))
                for (size_t i = i0; i < std::min(n, i0 + grain); ++i) body(i);
        });
    };
    for_lines([&](size_t i) { room[i + 1] = (int64_t)count_token_starts(buf + recs[i].colon + 1, buf + recs[i].e); });
    if (worker_error) std::rethrow_exception(worker_error);
    for (size_t i = 0; i < n; ++i) room[i + 1] += room[i];
    const size_t total_room = (size_t)room[n];
    lap("count");

    const size_t map_len = std::max<size_t>(total_room * 8, 8);
    void* base = ::mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return false;
    ::madvise(base, map_len, MADV_HUGEPAGE);
    uint64_t* const flat = reinterpret_cast<uint64_t*>(base);

    // the cache writers' shared state lives on the heap: the finishing thread may outlive this call
    struct CacheJob {
        std::string text, part;
        CsrHeader head{};
        int fd = -1;
        size_t head_bytes = 0, n = 0;
        const uint64_t* flat = nullptr;
        std::vector<int64_t> room;                             // copy: final offsets when no line falls short
        std::unique_ptr<std::atomic<int64_t>[]> got;           // -1 until line i is parsed, then its value count
        std::mutex claim;
        size_t next = 0;
        std::atomic<bool> stop{false}, bad{false};             // stop: give up quietly; bad: a short line or a write error
        std::mutex m;
        std::condition_variable cv;
        int parse_state = 0;                                   // 1: `out` is complete, 2: the parse failed
        std::vector<std::thread> helpers;
    };
    std::shared_ptr<CacheJob> job;
    uint64_t name_bytes = 0;
    if (with_names)
        for (size_t i = 0; i < n; ++i) name_bytes += recs[i].colon - recs[i].b;
    std::unique_ptr<std::atomic<int64_t>[]> got_owner(new std::atomic<int64_t>[n ? n : 1]);
    std::atomic<int64_t>* got = got_owner.get();
    for (size_t i = 0; i < n; ++i) got[i].store(-1, std::memory_order_relaxed);
    if (!cache_for.empty() && with_names && total_room) {
        job = std::make_shared<CacheJob>();
        job->text = cache_for;
        job->part = csr_cache_part_path(cache_for);
        remove_stale_cache_parts(cache_for);
        if (text_identity(cache_for, job->head.text_size, job->head.text_mtime_ns))
            job->fd = ::open(job->part.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (job->fd < 0) {
            job.reset();
        } else {
            job->head_bytes = sizeof(CsrHeader) + (n + 1) * 8 + n * 8 + (size_t)((name_bytes + 7) / 8 * 8);
            job->n = n;
            job->flat = flat;
            job->room = room;
            job->got = std::move(got_owner);
            auto writer = [](std::shared_ptr<CacheJob> j) {
                // claim the next run of finished lines (up to 32 MiB), copy it into the file at its final place
                while (!j->stop && !j->bad) {
                    size_t i0, i1;
                    {
                        std::unique_lock<std::mutex> lock(j->claim);
                        i0 = j->next;
                        if (i0 >= j->n) return;
                        int64_t g;
                        while ((g = j->got[i0].load(std::memory_order_acquire)) < 0) {
                            if (j->stop || j->bad) return;
                            std::this_thread::sleep_for(std::chrono::microseconds(200));
                        }
                        i1 = i0;
                        while (i1 < j->n && (g = j->got[i1].load(std::memory_order_acquire)) >= 0 &&
                               (j->room[i1] - j->room[i0]) * 8 < (32 << 20)) {
                            if (g != j->room[i1 + 1] - j->room[i1]) {          // a short line: every later offset moves
                                j->bad = true;
                                return;
                            }
                            ++i1;
                        }
                        j->next = i1;
                    }
                    const char* src = reinterpret_cast<const char*>(j->flat + j->room[i0]);
                    size_t left = (size_t)(j->room[i1] - j->room[i0]) * 8;
                    uint64_t off = j->head_bytes + (uint64_t)j->room[i0] * 8;
                    while (left) {
                        const ssize_t w = ::pwrite(j->fd, src, left, (off_t)off);
                        if (w <= 0) {
                            j->bad = true;
                            return;
                        }
                        src += w;
                        left -= (size_t)w;
                        off += (uint64_t)w;
                    }
                }
            };
            got = job->got.get();
            try {
                for (int t = 0; t < 3; ++t) job->helpers.emplace_back(writer, job);
            } catch (const std::exception&) {
                // no thread to be had: the parse goes on without the cache (a joinable thread must not be destroyed)
                job->stop = true;
                for (auto& th : job->helpers) th.join();
                job->helpers.clear();
                ::close(job->fd);
                ::unlink(job->part.c_str());
                got_owner = std::move(job->got);
                job.reset();
            }
        }
    }
    // ends the cache job on the way out of a failed parse
    auto abandon_cache = [&]() {
        if (!job) return;
        job->stop = true;
        for (auto& th : job->helpers) th.join();
        ::close(job->fd);
        ::unlink(job->part.c_str());
        job.reset();
    };

    // pass 2: parse every line in place; sort + unique where the text was not strictly increasing
    for_lines([&](size_t i) {
        bool increasing = true;
        uint64_t* dst = flat + room[i];
        size_t k = parse_u64_raw(buf + recs[i].colon + 1, buf + recs[i].e, buf, map_end, dst, increasing);
        if (!increasing) k = sort_unique_u64(dst, k);
        got[i].store((int64_t)k, std::memory_order_release);
    });
    if (worker_error) {
        abandon_cache();
        ::munmap(base, map_len);
        std::rethrow_exception(worker_error);
    }
    lap("tokens -> values");
    // lines that gave fewer values than their room (a bad token, duplicates): close the gaps (the cache writers have
    // stopped at the first of them and touch nothing from there on)
    out.offsets.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) out.offsets[i + 1] = out.offsets[i] + got[i].load(std::memory_order_relaxed);
    const size_t total = (size_t)out.offsets[n];
    if (total != total_room) {
        if (job) {
            job->bad = true;
            for (auto& th : job->helpers) th.join();          // nobody reads the flat array while it is compacted
            job->helpers.clear();
        }
        for (size_t i = 0; i < n; ++i) {
            const int64_t g = out.offsets[i + 1] - out.offsets[i];
            if (out.offsets[i] != room[i] && g) memmove(flat + out.offsets[i], flat + room[i], (size_t)g * 8);
        }
    }
    out.names.assign(n, std::string());
    if (with_names)
        for (size_t i = 0; i < n; ++i) out.names[i].assign(buf + recs[i].b, recs[i].colon - recs[i].b);
    out.hashes.adopt_mapping(base, map_len, flat, total);
    lap("offsets, names");
    if (job) {
        // the finisher: waits for the writers, then either completes the file (head + rename) or, if a line fell short or
        // a write failed, writes the cache from the finished sets
        const HashSets* sets = &out;
        std::thread finisher([job, sets, name_bytes]() {
            for (auto& th : job->helpers) th.join();
            bool ok = !job->bad && !job->stop;
            if (ok) {
                CsrHeader h = job->head;
                memcpy(h.magic, "MVSCSR01", 8);
                h.samples = sets->names.size();
                h.values = sets->hashes.size();
                h.name_bytes = name_bytes;
                std::string head;
                head.reserve(job->head_bytes);
                head.append(reinterpret_cast<const char*>(&h), sizeof h);
                head.append(reinterpret_cast<const char*>(sets->offsets.data()), sets->offsets.size() * 8);
                uint64_t at = 0;
                for (const std::string& nm : sets->names) {
                    at += nm.size();
                    head.append(reinterpret_cast<const char*>(&at), 8);
                }
                for (const std::string& nm : sets->names) head.append(nm);
                if (at % 8) head.append(8 - at % 8, '\0');
                ok = head.size() == job->head_bytes && ::pwrite(job->fd, head.data(), head.size(), 0) == (ssize_t)head.size();
            }
            ok = (::close(job->fd) == 0) && ok;
            if (ok && ::rename(job->part.c_str(), csr_cache_path(job->text).c_str()) == 0) return;
            ::unlink(job->part.c_str());
            if (job->bad && !job->stop) (void)write_csr_cache(job->text, *sets, 0);
        });
        if (cache_writer) *cache_writer = std::move(finisher);
        else finisher.join();
    }
    return true;
}

// best effort: a cache that cannot be written is simply not there next time.  The values (gigabytes: 4 GB for 10k samples
// of 50k hashes) are written by several threads at their offsets -- the copy into the page cache is what takes the time
// and one thread moves ~2-5 GB/s; callers that have better things to do run this on a thread of its own.
inline bool write_csr_cache(const std::string& hash_file, const HashSets& sets, unsigned threads) {
    CsrHeader h{};
    memcpy(h.magic, "MVSCSR01", 8);
    if (!text_identity(hash_file, h.text_size, h.text_mtime_ns)) return false;
    h.samples = sets.names.size();
    h.values = sets.hashes.size();
    std::vector<uint64_t> ends(sets.names.size());
    uint64_t at = 0;
    for (size_t i = 0; i < sets.names.size(); ++i) ends[i] = at += sets.names[i].size();
    h.name_bytes = at;
    const std::string path = csr_cache_path(hash_file), tmp = csr_cache_part_path(hash_file);
    remove_stale_cache_parts(hash_file);
    // head: header, offsets, name ends, names, padding to 8
    std::string head;
    head.reserve(sizeof h + sets.offsets.size() * 8 + ends.size() * 8 + (size_t)at + 8);
    head.append(reinterpret_cast<const char*>(&h), sizeof h);
    head.append(reinterpret_cast<const char*>(sets.offsets.data()), sets.offsets.size() * 8);
    head.append(reinterpret_cast<const char*>(ends.data()), ends.size() * 8);
    for (const std::string& nm : sets.names) head.append(nm);
    if (at % 8) head.append(8 - at % 8, '\0');
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    auto write_at = [fd](const char* p, size_t n, uint64_t off) {
        while (n) {
            const ssize_t w = ::pwrite(fd, p, std::min<size_t>(n, (size_t)1 << 30), (off_t)off);
            if (w <= 0) return false;
            p += w;
            n -= (size_t)w;
            off += (uint64_t)w;
        }
        return true;
    };
    std::atomic<bool> ok{write_at(head.data(), head.size(), 0)};
    const size_t value_bytes = (size_t)h.values * 8;
    if (ok && value_bytes) {
        if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const size_t chunk = (size_t)64 << 20;
        const size_t n_chunks = (value_bytes + chunk - 1) / chunk;
        threads = (unsigned)std::min<size_t>(threads, n_chunks);
        std::atomic<size_t> next{0};
        const char* base = reinterpret_cast<const char*>(sets.hashes.data());
        auto work = [&]() {
            for (size_t i = next++; i < n_chunks && ok; i = next++) {
                const size_t o = i * chunk, n = std::min(chunk, value_bytes - o);
                if (!write_at(base + o, n, head.size() + o)) ok = false;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    const bool closed = ::close(fd) == 0;
    if (!ok || !closed || rename(tmp.c_str(), path.c_str()) != 0) {
        ::unlink(tmp.c_str());
        return false;
    }
    return true;
}

// true: `out` holds the cached sets (values mapped, not copied).  false: no cache, stale cache, or damaged file.
inline bool load_csr_cache(const std::string& hash_file, HashSets& out) {
    uint64_t text_size = 0;
    int64_t text_mtime = 0;
    if (!text_identity(hash_file, text_size, text_mtime)) return false;
    const std::string path = csr_cache_path(hash_file);
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(CsrHeader)) {
        ::close(fd);
        return false;
    }
    const size_t len = (size_t)st.st_size;
    void* m = ::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return false;
    const char* base = (const char*)m;
    CsrHeader h;
    memcpy(&h, base, sizeof h);
    const size_t names_pad = (size_t)((h.name_bytes + 7) / 8 * 8);
    const bool sane = memcmp(h.magic, "MVSCSR01", 8) == 0 && h.text_size == text_size && h.text_mtime_ns == text_mtime &&
                      h.samples < (1ULL << 40) && h.values < (1ULL << 48) && h.name_bytes < (1ULL << 40) &&
                      len == sizeof h + (h.samples + 1) * 8 + h.samples * 8 + names_pad + h.values * 8;
    if (!sane) {
        ::munmap(m, len);
        return false;
    }
    const int64_t* offs = (const int64_t*)(base + sizeof h);
    const uint64_t* ends = (const uint64_t*)(offs + h.samples + 1);
    const char* names = (const char*)(ends + h.samples);
    const uint64_t* values = (const uint64_t*)(names + names_pad);
    bool ok = offs[0] == 0 && (uint64_t)offs[h.samples] == h.values && (h.samples == 0 || ends[h.samples - 1] == h.name_bytes);
    for (uint64_t i = 0; ok && i < h.samples; ++i)
        ok = offs[i + 1] >= offs[i] && (i == 0 || ends[i] >= ends[i - 1]);
    if (!ok) {
        ::munmap(m, len);
        return false;
    }
    out.offsets.assign(offs, offs + h.samples + 1);
    out.names.resize((size_t)h.samples);
    for (uint64_t i = 0; i < h.samples; ++i) {
        const uint64_t b = i ? ends[i - 1] : 0;
        out.names[(size_t)i].assign(names + b, (size_t)(ends[i] - b));
    }
    ::madvise(m, len, MADV_SEQUENTIAL);
    out.hashes.adopt_mapping(m, len, values, (size_t)h.values);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// DB folder (src/project_everything.cpp:306-361 writes it, src/pairwise_comp_optimized.cpp:852-914 reads it)
// ---------------------------------------------------------------------------------------------------
// `os << double` with default precision / flags == "%g"
inline std::string format_g(double v) {
    char b[64];
    snprintf(b, sizeof b, "%g", v);
    return b;
}
// `os << float` likewise (src/standalone_projection.cpp:40)
inline std::string format_g_float(float v) { return format_g((double)v); }

// norm of one sketch: the reference takes the float32 path (cast / sqrt(float d), Eigen norm) whose
// last digit is not reproducible; this build defines it as sqrt(double(sumsq) / d).
inline double norm_from_sumsq(int64_t sumsq, int d) { return std::sqrt((double)sumsq / (double)d); }
// the float32 evaluation of src/project_everything.cpp:328-329, summed in index order
inline float norm_float32_path(const int32_t* v, int d) {
    const float root = std::sqrt((float)d);
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) {
        const float f = (float)v[k] / root;
        acc += f * f;
    }
    return std::sqrt(acc);
}

struct DbInfo {
    std::string dtype = "int32";
    int dimension = 0;
    int64_t total_vectors = 0;
    std::vector<std::string> names;
    std::vector<double> norms_sq;   // stod(text)^2, src/pairwise_comp_optimized.cpp:893-901
};

inline bool read_norms(const std::string& norms_file, DbInfo& db) {
    std::ifstream in(norms_file);
    if (!in) return false;
    std::string line;
    while (std::getline(in, line)) {
        const size_t pos = line.find(' ');
        if (pos == std::string::npos) continue;
        const double norm = std::stod(line.substr(pos + 1));
        db.names.push_back(line.substr(0, pos));
        db.norms_sq.push_back(norm * norm);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// matrix shard folder, active format (writer src/pairwise_comp_optimized.cpp:645-817)
// ---------------------------------------------------------------------------------------------------
struct ShardStats {
    uint64_t jac_space = 0, ngh_space = 0, rows = 0;
};

// cells must be grouped by row with ascending columns inside a row (mvs_pairwise_rows order).
// threads: 0 = all host threads (rows are encoded independently).
// Rows are written in ascending order (the reference iterates a std::unordered_map, i.e. in an
// unspecified order; its reader builds a map and accepts any order).
inline ShardStats write_shard(const std::string& folder, const mvs_cell* cells, size_t n_cells,
                              unsigned threads = 0) {
    if (!fs::exists(folder)) fs::create_directories(folder);
    std::ofstream bin_out(folder + "matrix.bin", std::ios::binary);
    std::ofstream index_out(folder + "row_index.bin", std::ios::binary);
    // row table (first cell of every row), then the rows are encoded by workers, each a contiguous run of
    // rows into its own buffer; the buffers are written out in order, so the file is the one a single
    // thread would write
    std::vector<size_t> row_first;
    for (size_t i = 0; i < n_cells; ++i)
        if (i == 0 || cells[i].row != cells[i - 1].row) row_first.push_back(i);
    const size_t n_rows = row_first.size();
    row_first.push_back(n_cells);
    std::vector<uint32_t> row_vec(n_rows), start_neighbor(n_rows);
    std::vector<uint64_t> curr_pos_vec(n_rows);
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n_rows / 4096));
    threads = std::max(1u, threads);
    struct Part {
        std::string bytes;
        uint64_t jac_space = 0, ngh_space = 0;
        std::string error;
    };
    std::vector<Part> parts(threads);
    auto encode_part = [&](unsigned t) {
        Part& part = parts[t];
        std::ostringstream os(std::ios::binary);
        const size_t r0 = n_rows * t / threads, r1 = n_rows * (t + 1) / threads;
        for (size_t r = r0; r < r1; ++r) {
            const size_t i = row_first[r], j = row_first[r + 1];
            row_vec[r] = (uint32_t)cells[i].row;
            curr_pos_vec[r] = (uint64_t)os.tellp();             // relative to this part; rebased below
            start_neighbor[r] = (uint32_t)cells[i].col;
            std::vector<uint16_t> jac(j - i);
            std::vector<uint64_t> delta(j - i - 1);
            for (size_t k = i; k < j; ++k) {
                jac[k - i] = (uint16_t)cells[k].q;
                if (k > i) {
                    if (cells[k].col <= cells[k - 1].col) {
                        part.error = "write_shard: columns not ascending";
                        return;
                    }
                    delta[k - i - 1] = (uint64_t)(cells[k].col - cells[k - 1].col);
                }
            }
            mvs_codec::compact_vector cv_jc;
            cv_jc.build(jac.begin(), jac.size());
            cv_jc.save(os);
            part.jac_space += cv_jc.num_bytes();
            if (jac.size() > 1) {                     // :732 a single-entry row has no delta sequence
                mvs_codec::rice_sequence rs_delta;
                rs_delta.encode(delta.begin(), delta.size());
                rs_delta.save(os);
                part.ngh_space += rs_delta.num_bytes();
            }
        }
        part.bytes = os.str();
    };
    // an exception inside a worker (the codec's runtime_error, bad_alloc) travels through Part::error to the caller
    // instead of ending the process in std::terminate
    auto work = [&](unsigned t) {
        try {
            encode_part(t);
        } catch (const std::exception& e) {
            parts[t].error = std::string("write_shard: ") + e.what();
        } catch (...) {
            parts[t].error = "write_shard: unknown error in an encoder thread";
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    ShardStats st;
    uint64_t base = 0;
    for (unsigned t = 0; t < threads; ++t) {
        if (!parts[t].error.empty()) throw std::runtime_error(parts[t].error);
        const size_t r0 = n_rows * t / threads, r1 = n_rows * (t + 1) / threads;
        for (size_t r = r0; r < r1; ++r) curr_pos_vec[r] += base;
        bin_out.write(parts[t].bytes.data(), (std::streamsize)parts[t].bytes.size());
        base += parts[t].bytes.size();
        st.jac_space += parts[t].jac_space;
        st.ngh_space += parts[t].ngh_space;
    }
    bin_out.close();
    st.rows = row_vec.size();
    // row ids, then byte-offset deltas between consecutive rows (:769-783)
    mvs_codec::compact_vector cv_rows;
    cv_rows.build(row_vec.begin(), row_vec.size());
    cv_rows.save(index_out);
    std::vector<uint64_t> pos_delta(curr_pos_vec.empty() ? 0 : curr_pos_vec.size() - 1);
    for (size_t k = 1; k < curr_pos_vec.size(); ++k) pos_delta[k - 1] = curr_pos_vec[k] - curr_pos_vec[k - 1];
    mvs_codec::compact_vector cv_cps;
    cv_cps.build(pos_delta.begin(), pos_delta.size());
    cv_cps.save(index_out);
    index_out.close();
    std::ofstream ngh_out(folder + "neighbor_start.bin", std::ios::binary);
    mvs_codec::rice_sequence rs_start;
    rs_start.encode(start_neighbor.begin(), start_neighbor.size());
    rs_start.save(ngh_out);
    st.ngh_space += rs_start.num_bytes();
    return st;
}

// The same shard written piece by piece: add() takes CSR pieces of consecutive whole rows in ascending row order
// (mvs_row_block, what mvs_pairwise_stream delivers) and appends their rows to matrix.bin at once; only the per-row
// directory (row id, byte offset, first column) stays in memory until finish() writes row_index.bin and
// neighbor_start.bin.  The files are byte for byte what write_shard() writes for the same cells.
class ShardWriter {
public:
    // The three files are written under ".part" names and renamed in finish(): a comparison that fails half way (out of
    // memory, a device error, an aborted callback) leaves a shard that was there before untouched, as the reference does --
    // it only opens its output once all results exist (src/pairwise_comp_optimized.cpp:645-817 runs after the tile loop).
    explicit ShardWriter(const std::string& folder, unsigned threads = 0) : folder_(folder), threads_(threads) {
        if (!fs::exists(folder_)) fs::create_directories(folder_);
        bin_fd_ = ::open((folder_ + "matrix.bin.part").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (bin_fd_ < 0) throw std::runtime_error("ShardWriter: cannot create " + folder_ + "matrix.bin.part");
        if (threads_ == 0) threads_ = std::max(1u, std::thread::hardware_concurrency());
        // matrix.bin is written by a few threads of its own (pwrite at the position every run of bytes is known to have):
        // the encoded rows of a piece go to the file while the next piece is being encoded or downloaded, and one thread
        // copying into the page cache would be what bounds a dense shard (1.4 GB at ~3 GB/s)
        try {
            spawn_file_threads();
        } catch (...) {                    // a thread that could not start: the ones that did must not outlive the object
            stop_file_threads();
            ::close(bin_fd_);
            ::unlink((folder_ + "matrix.bin.part").c_str());
            throw;
        }
    }
    ShardWriter(const ShardWriter&) = delete;
    ShardWriter& operator=(const ShardWriter&) = delete;
    ~ShardWriter() {
        stop_file_threads();
        if (bin_fd_ >= 0) ::close(bin_fd_);
        if (!finished_)                    // finish() did not complete: nothing half-written stays behind
            for (const char* f : {"matrix.bin.part", "row_index.bin.part", "neighbor_start.bin.part"}) ::unlink((folder_ + f).c_str());
    }

private:
    void spawn_file_threads() {
        for (int t = 0; t < 4; ++t)
            file_threads_.emplace_back([this] {
                for (;;) {
                    Job job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return closing_ || !pending_.empty(); });
                        if (pending_.empty()) return;
                        job = std::move(pending_.front());
                        pending_.pop_front();
                    }
                    cv_.notify_all();
                    size_t done = 0;
                    while (done < job.bytes.size()) {
                        const ssize_t w = ::pwrite(bin_fd_, job.bytes.data() + done, job.bytes.size() - done, (off_t)(job.pos + done));
                        if (w <= 0) {
                            std::lock_guard<std::mutex> lk(mu_);
                            write_failed_ = true;
                            break;
                        }
                        done += (size_t)w;
                    }
                }
            });
    }

public:
    void add(const mvs_row_block& b) {
        const int64_t rows = b.row_end - b.row_begin;
        if (rows < 0 || b.row_begin < next_row_) throw std::runtime_error("ShardWriter: pieces must come in ascending row order");
        next_row_ = b.row_end;
        cells_ += (uint64_t)b.n_cells;
        std::vector<int64_t> live;                                  // rows of the piece that hold cells
        for (int64_t r = 0; r < rows; ++r)
            if (b.row_ptr[r + 1] > b.row_ptr[r]) live.push_back(r);
        const size_t n_rows = live.size();
        if (n_rows == 0) return;
        const size_t first = row_vec_.size();
        row_vec_.resize(first + n_rows);
        start_neighbor_.resize(first + n_rows);
        curr_pos_vec_.resize(first + n_rows);
        // encoder threads by the piece's CELLS (a dense piece is a few hundred rows of ten thousand cells each), every
        // thread a contiguous run of rows holding about the same number of cells
        unsigned threads = (unsigned)std::min<size_t>(threads_, std::max<size_t>(1, (size_t)b.n_cells / 32768));
        threads = (unsigned)std::min<size_t>(std::max(1u, threads), n_rows);
        std::vector<size_t> cut(threads + 1, n_rows);
        cut[0] = 0;
        for (unsigned t = 1; t < threads; ++t) {
            const int64_t target = b.n_cells / threads * t;
            size_t lo = cut[t - 1], hi = n_rows;
            while (lo < hi) {                                       // first live row whose first cell is at or past the target
                const size_t mid = (lo + hi) / 2;
                if (b.row_ptr[live[mid]] < target) lo = mid + 1;
                else hi = mid;
            }
            cut[t] = lo;
        }
        struct Part {
            std::string bytes;
            uint64_t jac_space = 0, ngh_space = 0;
            std::string error;
        };
        std::vector<Part> parts(threads);
        auto encode_part = [&](unsigned t) {
            Part& part = parts[t];
            std::ostringstream os(std::ios::binary);
            const size_t r0 = cut[t], r1 = cut[t + 1];
            std::vector<uint16_t> jac;
            std::vector<uint64_t> delta;
            for (size_t r = r0; r < r1; ++r) {
                const int64_t i = b.row_ptr[live[r]], j = b.row_ptr[live[r] + 1];
                row_vec_[first + r] = (uint32_t)(b.row_begin + live[r]);
                curr_pos_vec_[first + r] = (uint64_t)os.tellp();        // relative to this part; rebased below
                start_neighbor_[first + r] = (uint32_t)b.col[i];
                jac.resize((size_t)(j - i));
                delta.resize((size_t)(j - i - 1));
                for (int64_t k = i; k < j; ++k) {
                    jac[(size_t)(k - i)] = b.q ? (uint16_t)b.q[k] : b.q16[k];
                    if (k > i) {
                        if (b.col[k] <= b.col[k - 1]) {
                            part.error = "ShardWriter: columns not ascending";
                            return;
                        }
                        delta[(size_t)(k - i - 1)] = (uint64_t)(b.col[k] - b.col[k - 1]);
                    }
                }
                mvs_codec::compact_vector cv_jc;
                cv_jc.build(jac.begin(), jac.size());
                cv_jc.save(os);
                part.jac_space += cv_jc.num_bytes();
                if (jac.size() > 1) {                     // :732 a single-entry row has no delta sequence
                    mvs_codec::rice_sequence rs_delta;
                    rs_delta.encode(delta.begin(), delta.size());
                    rs_delta.save(os);
                    part.ngh_space += rs_delta.num_bytes();
                }
            }
            part.bytes = os.str();
        };
        auto work = [&](unsigned t) {
            try {
                encode_part(t);
            } catch (const std::exception& e) {
                parts[t].error = std::string("ShardWriter: ") + e.what();
            } catch (...) {
                parts[t].error = "ShardWriter: unknown error in an encoder thread";
            }
        };
        if (threads == 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, t);
            for (auto& th : pool) th.join();
        }
        for (unsigned t = 0; t < threads; ++t) {
            if (!parts[t].error.empty()) throw std::runtime_error(parts[t].error);
            const size_t r0 = cut[t], r1 = cut[t + 1];
            for (size_t r = r0; r < r1; ++r) curr_pos_vec_[first + r] += pos_;
            const uint64_t at = pos_;
            pos_ += parts[t].bytes.size();
            enqueue(at, std::move(parts[t].bytes));
            stats_.jac_space += parts[t].jac_space;
            stats_.ngh_space += parts[t].ngh_space;
        }
    }

    // The same for a piece whose rows the device has already encoded (mvs_pairwise_stream_encoded): the records go to the
    // file as they are, the directory entries are rebased to the file position.
    void add_encoded(const mvs_encoded_rows& b) {
        if (b.row_begin < next_row_) throw std::runtime_error("ShardWriter: pieces must come in ascending row order");
        next_row_ = b.row_end;
        cells_ += (uint64_t)b.n_cells;
        if (b.n_rows == 0) return;
        uint64_t jac = 0;
        for (int64_t r = 0; r < b.n_rows; ++r) {
            row_vec_.push_back(b.rows[r]);
            start_neighbor_.push_back(b.first_col[r]);
            curr_pos_vec_.push_back(pos_ + b.offset[r]);
            jac += b.jac_bytes[r];
        }
        stats_.jac_space += jac;
        stats_.ngh_space += (uint64_t)b.n_bytes - jac;
        // the pinned buffer is only valid during the call: the piece is copied once, in a few runs so that the file threads
        // share it
        const uint64_t at = pos_;
        pos_ += (uint64_t)b.n_bytes;
        const size_t run = 8u << 20;
        for (size_t o = 0; o < (size_t)b.n_bytes; o += run)
            enqueue(at + o, std::string(reinterpret_cast<const char*>(b.bytes) + o, std::min(run, (size_t)b.n_bytes - o)));
    }

    uint64_t cells() const { return cells_; }

    ShardStats finish() {
        stop_file_threads();
        if (bin_fd_ >= 0) ::close(bin_fd_);
        bin_fd_ = -1;
        if (write_failed_) throw std::runtime_error("ShardWriter: writing " + folder_ + "matrix.bin failed");
        std::ofstream index_out(folder_ + "row_index.bin.part", std::ios::binary);
        stats_.rows = row_vec_.size();
        mvs_codec::compact_vector cv_rows;                          // row ids, then byte-offset deltas (:769-783)
        cv_rows.build(row_vec_.begin(), row_vec_.size());
        cv_rows.save(index_out);
        std::vector<uint64_t> pos_delta(curr_pos_vec_.empty() ? 0 : curr_pos_vec_.size() - 1);
        for (size_t k = 1; k < curr_pos_vec_.size(); ++k) pos_delta[k - 1] = curr_pos_vec_[k] - curr_pos_vec_[k - 1];
        mvs_codec::compact_vector cv_cps;
        cv_cps.build(pos_delta.begin(), pos_delta.size());
        cv_cps.save(index_out);
        index_out.close();
        std::ofstream ngh_out(folder_ + "neighbor_start.bin.part", std::ios::binary);
        mvs_codec::rice_sequence rs_start;
        rs_start.encode(start_neighbor_.begin(), start_neighbor_.size());
        rs_start.save(ngh_out);
        ngh_out.close();
        if (!index_out || !ngh_out) throw std::runtime_error("ShardWriter: writing the index files of " + folder_ + " failed");
        stats_.ngh_space += rs_start.num_bytes();
        for (const char* f : {"matrix.bin", "row_index.bin", "neighbor_start.bin"})
            if (::rename((folder_ + f + ".part").c_str(), (folder_ + f).c_str()) != 0)
                throw std::runtime_error("ShardWriter: cannot rename " + folder_ + f + ".part");
        finished_ = true;
        return stats_;
    }

private:
    struct Job {
        uint64_t pos = 0;
        std::string bytes;
    };
    void enqueue(uint64_t pos, std::string&& bytes) {
        if (bytes.empty()) return;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return pending_.size() < 64; });
            pending_.push_back(Job{pos, std::move(bytes)});
        }
        cv_.notify_all();
    }
    void stop_file_threads() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            closing_ = true;
        }
        cv_.notify_all();
        for (auto& th : file_threads_)
            if (th.joinable()) th.join();
    }
    std::string folder_;
    unsigned threads_;
    int bin_fd_ = -1;
    std::vector<std::thread> file_threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> pending_;
    bool closing_ = false, write_failed_ = false, finished_ = false;
    uint64_t pos_ = 0, cells_ = 0;
    int64_t next_row_ = 0;
    std::vector<uint32_t> row_vec_, start_neighbor_;
    std::vector<uint64_t> curr_pos_vec_;
    ShardStats stats_;
};

// decode a shard folder back into (row, col, q) triples in file order -- what the reader's
// load_neighbors_for_rows_jaccard_wo_sort (src/read_pc_mat_cmp.cpp:597-671) reconstructs
inline bool read_shard(const std::string& folder, std::vector<mvs_cell>& out) {
    std::ifstream index_in(folder + "row_index.bin", std::ios::binary);
    std::ifstream bin_in(folder + "matrix.bin", std::ios::binary);
    std::ifstream ngh_in(folder + "neighbor_start.bin", std::ios::binary);
    if (!index_in || !bin_in || !ngh_in) return false;
    mvs_codec::compact_vector cv_rows, cv_cps;
    cv_rows.load(index_in);
    cv_cps.load(index_in);
    mvs_codec::rice_sequence rs_start;
    rs_start.load(ngh_in);
    uint64_t addr = 0;
    for (uint64_t r = 0; r < cv_rows.size(); ++r) {
        if (r > 0) addr += cv_cps.access(r - 1);
        bin_in.seekg((std::streamoff)addr);
        mvs_codec::compact_vector jac;
        jac.load(bin_in);
        std::vector<uint64_t> delta;
        if (jac.size() > 1) {
            mvs_codec::rice_sequence rs;
            rs.load(bin_in);
            rs.decode(delta);
        }
        uint64_t col = rs_start.access(r);
        for (uint64_t k = 0; k < jac.size(); ++k) {
            if (k > 0) col += delta[k - 1];
            mvs_cell c;
            c.row = (int32_t)cv_rows.access(r);
            c.col = (int32_t)col;
            c.dot = 0;
            c.q = (int32_t)jac.access(k);
            out.push_back(c);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// legacy int16 shard format (writer src/pairwise_comp_optimized_16bits.cpp:251-323; the reference has no reader
// for it).  matrix.bin: per row elias_fano(cols, universe = last col + 1) then compact_vector(round(dot / d));
// row_index.bin: compact_vector(rows) + compact_vector(byte offset of each row in matrix.bin); both files are
// then compressed with zstd and the originals removed (:317-322, `system("zstd -f ...")`).  zstd is bound at run
// time (libzstd.so.1); where it is missing the uncompressed files stay, which is also what the reference's
// `zstd -f x && rm -f x` leaves behind when the command fails.
// ---------------------------------------------------------------------------------------------------
struct Zstd {
    size_t (*bound)(size_t) = nullptr;
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
    size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
    unsigned long long (*content_size)(const void*, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    bool ok = false;
    Zstd() {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        bound = reinterpret_cast<decltype(bound)>(dlsym(h, "ZSTD_compressBound"));
        compress = reinterpret_cast<decltype(compress)>(dlsym(h, "ZSTD_compress"));
        decompress = reinterpret_cast<decltype(decompress)>(dlsym(h, "ZSTD_decompress"));
        content_size = reinterpret_cast<decltype(content_size)>(dlsym(h, "ZSTD_getFrameContentSize"));
        is_error = reinterpret_cast<decltype(is_error)>(dlsym(h, "ZSTD_isError"));
        ok = bound && compress && decompress && content_size && is_error;
    }
    static const Zstd& get() {
        static Zstd z;
        return z;
    }
};

inline bool read_whole_file(const std::string& path, std::string& bytes) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    bytes.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    return true;
}

// `zstd -f path && rm -f path` (level 3, the tool's default)
inline bool zstd_file_and_remove(const std::string& path) {
    const Zstd& z = Zstd::get();
    std::string raw;
    if (!z.ok || !read_whole_file(path, raw)) return false;
    std::string out(z.bound(raw.size()), '\0');
    const size_t n = z.compress(&out[0], out.size(), raw.data(), raw.size(), 3);
    if (z.is_error(n)) return false;
    std::ofstream os(path + ".zst", std::ios::binary);
    os.write(out.data(), (std::streamsize)n);
    os.close();
    if (!os) return false;
    fs::remove(path);
    return true;
}

// contents of path, or of path.zst when only that exists
inline bool read_maybe_zstd(const std::string& path, std::string& bytes) {
    if (read_whole_file(path, bytes)) return true;
    std::string packed;
    const Zstd& z = Zstd::get();
    if (!z.ok || !read_whole_file(path + ".zst", packed)) return false;
    const unsigned long long n = z.content_size(packed.data(), packed.size());
    if (n > (1ULL << 40)) return false;                       // unknown / error sentinel values are huge
    bytes.assign((size_t)n, '\0');
    const size_t got = z.decompress(&bytes[0], bytes.size(), packed.data(), packed.size());
    return !z.is_error(got) && got == n;
}

// cells grouped by row with ascending columns (mvs_pairwise_rows order); returns the number of rows written
inline uint64_t write_shard_legacy16(const std::string& folder, const mvs_cell* cells, size_t n_cells, int dimension) {
    if (!fs::exists(folder)) fs::create_directories(folder);
    const std::string bin_filename = folder + "matrix.bin", index_filename = folder + "row_index.bin";
    std::ofstream bin_out(bin_filename, std::ios::binary);
    std::ofstream index_out(index_filename, std::ios::binary);
    std::vector<int32_t> row_vec;
    std::vector<int64_t> curr_pos_vec;
    int64_t current_pos = 0;
    for (size_t i = 0; i < n_cells;) {
        size_t j = i;
        while (j < n_cells && cells[j].row == cells[i].row) ++j;
        std::vector<int32_t> cols(j - i);
        std::vector<int64_t> vals(j - i);
        for (size_t k = i; k < j; ++k) {
            cols[k - i] = cells[k].col;
            vals[k - i] = (int64_t)std::round((double)cells[k].dot / (double)dimension);        // :274-279
        }
        row_vec.push_back(cells[i].row);
        curr_pos_vec.push_back(current_pos);
        mvs_codec::elias_fano ef;
        ef.encode(cols.begin(), cols.size(), (uint64_t)cols.back() + 1);                      // :294-296
        ef.save(bin_out);
        current_pos += (int64_t)ef.num_bytes();
        mvs_codec::compact_vector cv;
        cv.build(vals.begin(), vals.size());                                                   // :299-302
        cv.save(bin_out);
        current_pos += (int64_t)cv.num_bytes();
        i = j;
    }
    bin_out.close();
    mvs_codec::compact_vector cv_rows, cv_pos;
    cv_rows.build(row_vec.begin(), row_vec.size());
    cv_rows.save(index_out);
    cv_pos.build(curr_pos_vec.begin(), curr_pos_vec.size());
    cv_pos.save(index_out);
    index_out.close();
    if (!zstd_file_and_remove(bin_filename) || !zstd_file_and_remove(index_filename))
        std::cerr << "zstd not available: " << bin_filename << " / " << index_filename << " left uncompressed" << std::endl;
    return row_vec.size();
}

// decode the legacy int16 shard back into (row, col, value = round(dot / d)) triples (value in mvs_cell::dot, q = 0)
inline bool read_shard_legacy16(const std::string& folder, std::vector<mvs_cell>& out) {
    std::string bin, index;
    if (!read_maybe_zstd(folder + "matrix.bin", bin) || !read_maybe_zstd(folder + "row_index.bin", index)) return false;
    std::istringstream index_in(index, std::ios::binary), bin_in(bin, std::ios::binary);
    mvs_codec::compact_vector cv_rows, cv_pos;
    cv_rows.load(index_in);
    cv_pos.load(index_in);
    for (uint64_t r = 0; r < cv_rows.size(); ++r) {
        bin_in.seekg((std::streamoff)cv_pos.access(r));
        mvs_codec::elias_fano ef;
        ef.load(bin_in);
        mvs_codec::compact_vector cv;
        cv.load(bin_in);
        if (ef.size() != cv.size()) throw std::runtime_error("legacy shard: row lengths disagree");
        std::vector<uint64_t> cols;
        ef.decode(cols);
        for (uint64_t k = 0; k < cols.size(); ++k) {
            mvs_cell c;
            c.row = (int32_t)cv_rows.access(r);
            c.col = (int32_t)cols[k];
            c.dot = (int32_t)cv.access(k);
            c.q = 0;
            out.push_back(c);
        }
    }
    return true;
}

inline int pick_device() {
    const char* e = getenv("MVS_DEVICE");
    return e ? atoi(e) : 0;
}

}  // namespace mvs_host

#endif
