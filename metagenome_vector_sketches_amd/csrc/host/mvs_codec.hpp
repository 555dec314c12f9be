// mvs_codec.hpp -- succinct containers for the matrix shard files (matrix.bin, row_index.bin,
// neighbor_start.bin).
//
// The reference writes these files through `bits::compact_vector` and `bits::rice_sequence<>`
// (src/pairwise_comp_optimized.cpp:724-736, :769-790) from the `bits` git submodule
// (github.com/hasin-abrar/bits), which is ABSENT from the reference tree: its byte layout cannot be
// recovered, so byte compatibility with reference-built files is UNPINNED.  These
// classes keep the API shape the call sites use (build/encode, save, load, size, access, num_bytes)
// with a byte layout of our own, documented here, so that swapping in the real library is a
// one-include change.
//
// Layout (all little-endian u64 words)
//   compact_vector : [size][width][n_words][words...]            value i at bit i*width
//   elias_fano     : see the class (legacy int16 shard format only)
//   rice_sequence  : [size][k][low: compact_vector of width k (absent when k == 0)]
//                    [n_high_bits][n_words][high words...]       unary quotients: q zeros then a one
//                    [n_samples][sample...]                      bit position after every 64th terminator
#ifndef MVS_CODEC_HPP
#define MVS_CODEC_HPP

#include <cstdint>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace mvs_codec {

inline void put_u64(std::ostream& os, uint64_t v) { os.write(reinterpret_cast<const char*>(&v), 8); }
inline uint64_t get_u64(std::istream& is) {
    uint64_t v = 0;
    is.read(reinterpret_cast<char*>(&v), 8);
    if (!is) throw std::runtime_error("mvs_codec: truncated stream");
    return v;
}
inline void put_words(std::ostream& os, const std::vector<uint64_t>& w) {
    put_u64(os, w.size());
    if (!w.empty()) os.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)(w.size() * 8));
}
inline void get_words(std::istream& is, std::vector<uint64_t>& w) {
    const uint64_t n = get_u64(is);
    if (n > (1ULL << 40)) throw std::runtime_error("mvs_codec: implausible length");
    w.resize(n);
    if (n) is.read(reinterpret_cast<char*>(w.data()), (std::streamsize)(n * 8));
    if (!is) throw std::runtime_error("mvs_codec: truncated stream");
}
inline uint64_t bit_width(uint64_t v) {
    uint64_t w = 0;
    while (v) {
        ++w;
        v >>= 1;
    }
    return w;
}

class compact_vector {
public:
    template <typename It>
    void build(It begin, uint64_t n) {
        uint64_t mx = 0;
        It it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) {
            const uint64_t v = (uint64_t)*it;
            if (v > mx) mx = v;
        }
        build(begin, n, bit_width(mx) ? bit_width(mx) : 1);
    }
    template <typename It>
    void build(It begin, uint64_t n, uint64_t width) {
        if (width == 0 || width > 64) throw std::invalid_argument("compact_vector: width");
        m_size = n;
        m_width = width;
        m_data.assign((n * width + 63) / 64, 0);
        It it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) set(i, (uint64_t)*it);
    }
    // for a producer that computes its values on the fly: prepare(n, width), then put(i, v) for every i exactly once
    void prepare(uint64_t n, uint64_t width) {
        if (width == 0 || width > 64) throw std::invalid_argument("compact_vector: width");
        m_size = n;
        m_width = width;
        m_data.assign((n * width + 63) / 64, 0);
    }
    void put(uint64_t i, uint64_t v) { set(i, v); }
    uint64_t size() const { return m_size; }
    uint64_t width() const { return m_width; }
    uint64_t access(uint64_t i) const {
        const uint64_t pos = i * m_width, w = pos >> 6, off = pos & 63;
        uint64_t v = m_data[w] >> off;
        if (off + m_width > 64) v |= m_data[w + 1] << (64 - off);
        return m_width == 64 ? v : (v & ((1ULL << m_width) - 1));
    }
    uint64_t operator[](uint64_t i) const { return access(i); }
    uint64_t num_bytes() const { return 8 * (3 + m_data.size()); }
    void save(std::ostream& os) const {
        put_u64(os, m_size);
        put_u64(os, m_width);
        put_words(os, m_data);
    }
    void load(std::istream& is) {
        m_size = get_u64(is);
        m_width = get_u64(is);
        get_words(is, m_data);
        if (m_width == 0 || m_width > 64 || m_data.size() != (m_size * m_width + 63) / 64)
            throw std::runtime_error("compact_vector: corrupt header");
    }

private:
    void set(uint64_t i, uint64_t v) {
        const uint64_t pos = i * m_width, w = pos >> 6, off = pos & 63;
        m_data[w] |= v << off;
        if (off + m_width > 64) m_data[w + 1] |= v >> (64 - off);
    }
    uint64_t m_size = 0, m_width = 1;
    std::vector<uint64_t> m_data;
};

// Golomb-Rice coded sequence of arbitrary (unsorted) unsigned values with random access.
class rice_sequence {
public:
    template <typename It>
    void encode(It begin, uint64_t n) {
        m_size = n;
        // k = floor(log2(mean)): the classic choice, within half a bit of optimal for geometric data
        unsigned __int128 sum = 0;                            // exact (the mean's floor is what fixes k)
        It it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) sum += (uint64_t)*it;
        const uint64_t mean = n ? (uint64_t)(sum / n) : 0;
        m_k = mean > 1 ? bit_width(mean) - 1 : 0;
        if (m_k > 48) m_k = 48;
        // the unary part holds sum(v >> k) zeros and n terminators: sized once, the low bits go straight into their vector
        uint64_t high_bits = n;
        it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) high_bits += m_k ? ((uint64_t)*it >> m_k) : (uint64_t)*it;
        m_high.assign((high_bits + 63) / 64, 0);
        m_samples.clear();
        m_samples.reserve((n + 63) / 64);
        if (m_k) m_low.prepare(n, m_k);
        const uint64_t low_mask = m_k ? ((1ULL << m_k) - 1) : 0;
        uint64_t bitpos = 0;
        it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) {
            const uint64_t v = (uint64_t)*it;
            if ((i & 63) == 0) m_samples.push_back(bitpos);
            bitpos += m_k ? (v >> m_k) : v;                   // q zeros
            m_high[bitpos >> 6] |= 1ULL << (bitpos & 63);     // terminator
            ++bitpos;
            if (m_k) m_low.put(i, v & low_mask);
        }
        m_high_bits = bitpos;
    }
    uint64_t size() const { return m_size; }
    uint64_t access(uint64_t i) const {
        uint64_t pos = m_samples[i >> 6];
        uint64_t q = 0;
        for (uint64_t j = (i >> 6) << 6; j <= i; ++j) q = next_unary(pos);
        return (q << m_k) | (m_k ? m_low.access(i) : 0);
    }
    // sequential decode of all values
    void decode(std::vector<uint64_t>& out) const {
        out.resize(m_size);
        uint64_t pos = 0;
        for (uint64_t i = 0; i < m_size; ++i) {
            const uint64_t q = next_unary(pos);
            out[i] = (q << m_k) | (m_k ? m_low.access(i) : 0);
        }
    }
    uint64_t num_bytes() const {
        return 8 * (2 + 2 + m_high.size() + 1 + m_samples.size()) + (m_k ? m_low.num_bytes() : 0);
    }
    void save(std::ostream& os) const {
        put_u64(os, m_size);
        put_u64(os, m_k);
        if (m_k) m_low.save(os);
        put_u64(os, m_high_bits);
        put_words(os, m_high);
        put_words(os, m_samples);
    }
    void load(std::istream& is) {
        m_size = get_u64(is);
        m_k = get_u64(is);
        if (m_k > 48) throw std::runtime_error("rice_sequence: corrupt header");
        if (m_k) m_low.load(is);
        m_high_bits = get_u64(is);
        get_words(is, m_high);
        get_words(is, m_samples);
        if (m_samples.size() != (m_size + 63) / 64) throw std::runtime_error("rice_sequence: corrupt samples");
    }

private:
    // number of zeros before the next one at/after bit `pos`; advances pos past the one
    uint64_t next_unary(uint64_t& pos) const {
        uint64_t q = 0;
        uint64_t w = pos >> 6;
        uint64_t cur = m_high[w] >> (pos & 63);
        uint64_t avail = 64 - (pos & 63);
        while (cur == 0) {
            q += avail;
            ++w;
            cur = m_high[w];
            avail = 64;
        }
        const uint64_t tz = (uint64_t)__builtin_ctzll(cur);
        q += tz;
        pos += q + 1;
        return q;
    }
    uint64_t m_size = 0, m_k = 0, m_high_bits = 0;
    compact_vector m_low;
    std::vector<uint64_t> m_high, m_samples;
};

// Elias-Fano coded non-decreasing sequence below a universe bound, with random access.  The legacy int16 shard
// format stores a row's column list this way (src/pairwise_comp_optimized_16bits.cpp:294-297:
// `ef.encode(cols.begin(), cols.size(), cols.back() + 1)`).
// Layout: [size][universe][l][low: compact_vector of width l (absent when l == 0)]
//         [n_high_bits][n_words][high words...]   per element: (high part - previous high part) zeros, then a one
//         [n_samples][sample...]                  bit position after every 64th terminator
class elias_fano {
public:
    template <typename It>
    void encode(It begin, uint64_t n, uint64_t universe) {
        m_size = n;
        m_universe = universe;
        m_l = (n && universe / n > 1) ? bit_width(universe / n) - 1 : 0;
        std::vector<uint64_t> lows;
        lows.reserve(n);
        m_high.clear();
        m_samples.clear();
        uint64_t bitpos = 0, prev_high = 0, prev = 0;
        It it = begin;
        for (uint64_t i = 0; i < n; ++i, ++it) {
            const uint64_t v = (uint64_t)*it;
            if (v < prev || v >= universe) throw std::invalid_argument("elias_fano: sequence must ascend below the universe");
            prev = v;
            if ((i & 63) == 0) m_samples.push_back(bitpos);
            const uint64_t h = m_l ? (v >> m_l) : v;
            lows.push_back(m_l ? (v & ((1ULL << m_l) - 1)) : 0);
            bitpos += h - prev_high;
            prev_high = h;
            const uint64_t w = bitpos >> 6;
            if (m_high.size() <= w) m_high.resize(w + 1, 0);
            m_high[w] |= 1ULL << (bitpos & 63);
            ++bitpos;
        }
        m_high_bits = bitpos;
        if (m_l) m_low.build(lows.begin(), n, m_l);
    }
    uint64_t size() const { return m_size; }
    uint64_t universe() const { return m_universe; }
    void decode(std::vector<uint64_t>& out) const {
        out.resize(m_size);
        uint64_t pos = 0, high = 0;
        for (uint64_t i = 0; i < m_size; ++i) {
            high += next_unary(pos);
            out[i] = (high << m_l) | (m_l ? m_low.access(i) : 0);
        }
    }
    uint64_t access(uint64_t i) const {
        // the high part of element i = (position of its terminator) - i
        uint64_t pos = m_samples[i >> 6];
        for (uint64_t j = (i >> 6) << 6; j <= i; ++j) next_unary(pos);
        const uint64_t high = pos - 1 - i;
        return (high << m_l) | (m_l ? m_low.access(i) : 0);
    }
    uint64_t num_bytes() const {
        return 8 * (3 + 2 + m_high.size() + 1 + m_samples.size()) + (m_l ? m_low.num_bytes() : 0);
    }
    void save(std::ostream& os) const {
        put_u64(os, m_size);
        put_u64(os, m_universe);
        put_u64(os, m_l);
        if (m_l) m_low.save(os);
        put_u64(os, m_high_bits);
        put_words(os, m_high);
        put_words(os, m_samples);
    }
    void load(std::istream& is) {
        m_size = get_u64(is);
        m_universe = get_u64(is);
        m_l = get_u64(is);
        if (m_l > 63) throw std::runtime_error("elias_fano: corrupt header");
        if (m_l) m_low.load(is);
        m_high_bits = get_u64(is);
        get_words(is, m_high);
        get_words(is, m_samples);
        if (m_samples.size() != (m_size + 63) / 64) throw std::runtime_error("elias_fano: corrupt samples");
    }

private:
    uint64_t next_unary(uint64_t& pos) const {
        uint64_t q = 0;
        uint64_t w = pos >> 6;
        uint64_t cur = m_high[w] >> (pos & 63);
        uint64_t avail = 64 - (pos & 63);
        while (cur == 0) {
            q += avail;
            ++w;
            cur = m_high[w];
            avail = 64;
        }
        q += (uint64_t)__builtin_ctzll(cur);
        pos += q + 1;
        return q;
    }
    uint64_t m_size = 0, m_universe = 0, m_l = 0, m_high_bits = 0;
    compact_vector m_low;
    std::vector<uint64_t> m_high, m_samples;
};

}  // namespace mvs_codec

#endif
