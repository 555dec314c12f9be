// mvs_ingest.hpp -- `project_everything convert`: sourmash .sig.zip archives -> "name: h1 h2 ..." text
// (reference: src/project_everything.cpp:73-235).  The reference shells out to `unzip` and `gunzip`
// (system(), /tmp/signature_extract<tid>); here the zip central directory is walked and the members are
// inflated in-process with zlib.  Same selection rule: every signatures/*.gz member whose JSON has
// "ksize": 31, hashes = the first "mins" array of that member (:109-150); sample name = file name up to
// the first '.' (:210-211).  Host-only code, outside the accelerated path.
#ifndef MVS_INGEST_HPP
#define MVS_INGEST_HPP

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace mvs_ingest {

inline uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// inflate `in` (raw deflate when window_bits < 0, gzip when 16 + MAX_WBITS)
inline bool inflate_all(const unsigned char* in, size_t n, int window_bits, std::string& out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, window_bits) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char*>(in);
    zs.avail_in = (uInt)n;
    out.clear();
    char buf[1 << 16];
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        zs.next_out = reinterpret_cast<unsigned char*>(buf);
        zs.avail_out = sizeof buf;
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) {
            inflateEnd(&zs);
            return false;
        }
        out.append(buf, sizeof buf - zs.avail_out);
        if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) break;   // truncated input
    }
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}

struct ZipMember {
    std::string name;
    std::string data;   // uncompressed
};

// all members of a (non-zip64) archive held in memory
inline bool read_zip(const std::string& path, std::vector<ZipMember>& members) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamoff size = f.tellg();
    if (size < 22) return false;
    std::string buf((size_t)size, '\0');
    f.seekg(0);
    f.read(&buf[0], size);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(buf.data());
    // end of central directory: signature 0x06054b50, searched backwards over the comment field
    size_t eocd = std::string::npos;
    for (size_t i = buf.size() - 22 + 1; i-- > 0;) {
        if (rd32(b + i) == 0x06054b50u) {
            eocd = i;
            break;
        }
        if (buf.size() - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) return false;
    const uint16_t n_entries = rd16(b + eocd + 10);
    size_t p = rd32(b + eocd + 16);
    for (uint16_t e = 0; e < n_entries; ++e) {
        if (p + 46 > buf.size() || rd32(b + p) != 0x02014b50u) return false;
        const uint16_t method = rd16(b + p + 10);
        const uint32_t csize = rd32(b + p + 20), usize = rd32(b + p + 24);
        const uint16_t nlen = rd16(b + p + 28), xlen = rd16(b + p + 30), clen = rd16(b + p + 32);
        const uint32_t lho = rd32(b + p + 42);
        ZipMember m;
        m.name.assign(buf, p + 46, nlen);
        if ((size_t)lho + 30 > buf.size() || rd32(b + lho) != 0x04034b50u) return false;
        const size_t data = (size_t)lho + 30 + rd16(b + lho + 26) + rd16(b + lho + 28);
        if (data + csize > buf.size()) return false;
        if (method == 0) {
            m.data.assign(buf, data, csize);
        } else if (method == 8) {
            if (!inflate_all(b + data, csize, -MAX_WBITS, m.data)) return false;
        } else {
            return false;
        }
        (void)usize;
        members.push_back(std::move(m));
        p += 46 + (size_t)nlen + xlen + clen;
    }
    return true;
}

inline bool ends_with(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// src/project_everything.cpp:109-150: hand-rolled scrape, first "ksize" must be 31, first "mins" array
inline void scrape_mins(const std::string& json_str, std::vector<uint64_t>& hashes) {
    const size_t ksize_pos = json_str.find("\"ksize\"");
    if (ksize_pos == std::string::npos) return;
    const size_t colon_pos = json_str.find(':', ksize_pos);
    if (colon_pos == std::string::npos) return;
    const size_t ksize_end = json_str.find_first_of(",}", colon_pos);
    std::string ksize_str = json_str.substr(colon_pos + 1, ksize_end - colon_pos - 1);
    ksize_str.erase(std::remove_if(ksize_str.begin(), ksize_str.end(), ::isspace), ksize_str.end());
    if (ksize_str != "31") return;
    const size_t mins_pos = json_str.find("\"mins\"");
    if (mins_pos == std::string::npos) return;
    const size_t array_start = json_str.find('[', mins_pos);
    const size_t array_end = json_str.find(']', array_start);
    if (array_start == std::string::npos || array_end == std::string::npos) return;
    size_t pos = array_start + 1;
    while (pos < array_end) {
        while (pos < array_end && (std::isspace((unsigned char)json_str[pos]) || json_str[pos] == ',')) ++pos;
        if (pos >= array_end) break;
        uint64_t v = 0;
        bool any = false, overflow = false;
        while (pos < array_end && json_str[pos] >= '0' && json_str[pos] <= '9') {
            const uint64_t dgt = (uint64_t)(json_str[pos] - '0');
            if (v > (UINT64_MAX - dgt) / 10) overflow = true;
            v = v * 10 + dgt;
            any = true;
            ++pos;
        }
        if (any && !overflow) hashes.push_back(v);                       // parse errors are ignored (:145-147)
        while (pos < array_end && json_str[pos] != ',') ++pos;           // skip the rest of a malformed token
    }
}

// one .sig.zip -> sorted unique hashes of its k = 31 signatures (:94-153)
inline bool load_signatures(const std::string& zip_path, std::vector<uint64_t>& hashes) {
    std::vector<ZipMember> members;
    if (!read_zip(zip_path, members)) return false;
    for (const ZipMember& m : members) {
        if (m.name.rfind("signatures/", 0) != 0 || !ends_with(m.name, ".gz")) continue;
        std::string json;
        if (!inflate_all(reinterpret_cast<const unsigned char*>(m.data.data()), m.data.size(), 16 + MAX_WBITS, json))
            continue;                                                    // "Error running gunzip": member skipped (:76-80)
        scrape_mins(json, hashes);
    }
    std::sort(hashes.begin(), hashes.end());
    hashes.erase(std::unique(hashes.begin(), hashes.end()), hashes.end());
    return true;
}

}  // namespace mvs_ingest

#endif
