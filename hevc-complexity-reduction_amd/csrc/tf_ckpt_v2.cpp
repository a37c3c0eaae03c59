// tf_ckpt_v2.cpp -- reader for TensorFlow "V2" checkpoint bundles (tf.train.Saver
// write_version=V2), the format of the reference's model files
// (/root/reference/HM-16.5_Test_AI/bin/video_to_cu_depth.py:29,126-133 -> saver.restore).
//
//   <prefix>.index                 LevelDB-style sorted table:
//        footer (48 B) = metaindex handle, index handle (varint64 offset,size), pad, magic
//        block = entries + restart array + n_restarts ; trailer = 1 B type + 4 B crc
//        entry = varint32 shared | varint32 non_shared | varint32 value_len | key delta | value
//        key ""  -> BundleHeaderProto ; other keys -> BundleEntryProto
//   <prefix>.data-00000-of-00001   raw little-endian tensors back to back
//
// TensorFlow is not a dependency: the table and the two protobuf messages are decoded here.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ethcnn_spec.h"

namespace ethcnn {

// ------------------------------------------------------------------------ crc32c ----
// slicing-by-8 tables, built once: a function-local static is initialised exactly once under concurrent first calls (C++11),
// which a hand-rolled "if (!done)" flag is not -- ethcnn_load_checkpoint can be called from several threads on several contexts
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};
uint32_t crc32c(const void* data, size_t n) {
    static const CrcTables tables;
    const uint32_t (&g_crc_table)[8][256] = tables.t;
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4);
        std::memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_table[7][lo & 0xff] ^ g_crc_table[6][(lo >> 8) & 0xff] ^ g_crc_table[5][(lo >> 16) & 0xff] ^
            g_crc_table[4][lo >> 24] ^ g_crc_table[3][hi & 0xff] ^ g_crc_table[2][(hi >> 8) & 0xff] ^
            g_crc_table[1][(hi >> 16) & 0xff] ^ g_crc_table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

// ------------------------------------------------------------------ byte cursor -----
struct Cur {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    uint32_t fixed32() {
        if (end - p < 4) { ok = false; return 0; }
        uint32_t v;
        std::memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    Cur sub(size_t n) {
        if ((size_t)(end - p) < n) { ok = false; return Cur{p, p}; }
        Cur c{p, p + n};
        p += n;
        return c;
    }
    void skip_field(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: sub(8); break;
            case 2: { const uint64_t n = varint(); sub((size_t)n); break; }
            case 5: sub(4); break;
            default: ok = false;
        }
    }
};

static int fail(char* err, size_t cap, int code, const std::string& msg) {
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
    return code;
}

static bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    const size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == (size_t)n;
}

// block at [off, off+size) followed by 1-byte type + 4-byte masked crc
static bool get_block(const std::vector<uint8_t>& file, uint64_t off, uint64_t size, Cur& out, std::string& why) {
    // overflow-safe: off and size are varints from the file
    if (off > file.size() || size > file.size() - off || file.size() - off - size < 5) { why = "block handle out of range"; return false; }
    const uint8_t* b = file.data() + off;
    if (b[size] != 0) { why = "compressed index block (unsupported)"; return false; }
    uint32_t stored;
    std::memcpy(&stored, b + size + 1, 4);
    if (crc32c_mask(crc32c(b, size + 1)) != stored) { why = "index block crc mismatch"; return false; }
    out = Cur{b, b + size};
    return true;
}

// iterate the entries of one block, calling fn(key, value cursor)
template <typename Fn>
static bool for_each_entry(Cur blk, Fn fn, std::string& why) {
    if (blk.end - blk.p < 4) { why = "short block"; return false; }
    uint32_t nrestart;
    std::memcpy(&nrestart, blk.end - 4, 4);
    if ((uint64_t)(blk.end - blk.p) < 4ull + 4ull * nrestart) { why = "bad restart array"; return false; }
    Cur c{blk.p, blk.end - 4 - 4 * (size_t)nrestart};
    std::string key;
    while (c.p < c.end) {
        const uint64_t shared = c.varint(), non_shared = c.varint(), vlen = c.varint();
        if (!c.ok || shared > key.size()) { why = "bad entry header"; return false; }
        Cur kd = c.sub((size_t)non_shared);
        Cur val = c.sub((size_t)vlen);
        if (!c.ok) { why = "entry overruns block"; return false; }
        key.resize((size_t)shared);
        key.append((const char*)kd.p, (size_t)non_shared);
        if (!fn(key, val)) {
            if (why.empty()) why = "bad entry value for key '" + key + "'";  // keep a nested, more specific reason
            return false;
        }
    }
    return true;
}

static bool parse_handle(Cur& c, uint64_t& off, uint64_t& size) {
    off = c.varint();
    size = c.varint();
    return c.ok;
}

static bool parse_entry_proto(Cur v, CkptEntry& e) {
    e.dtype = 0; e.rank = 0; e.shard = 0; e.offset = 0; e.size = 0; e.crc32c = 0;
    for (int i = 0; i < 4; ++i) e.shape[i] = 0;
    while (v.p < v.end && v.ok) {
        const uint64_t tag = v.varint();
        const int field = (int)(tag >> 3), wire = (int)(tag & 7);
        if (field == 1 && wire == 0) e.dtype = (int)v.varint();
        else if (field == 2 && wire == 2) {  // TensorShapeProto
            Cur s = v.sub((size_t)v.varint());
            while (s.p < s.end && s.ok) {
                const uint64_t t2 = s.varint();
                if ((t2 >> 3) == 2 && (t2 & 7) == 2) {  // Dim
                    Cur d = s.sub((size_t)s.varint());
                    int64_t sz = 0;
                    while (d.p < d.end && d.ok) {
                        const uint64_t t3 = d.varint();
                        if ((t3 >> 3) == 1 && (t3 & 7) == 0) sz = (int64_t)d.varint();
                        else d.skip_field((int)(t3 & 7));
                    }
                    if (!d.ok || e.rank >= 4) return false;
                    e.shape[e.rank++] = sz;
                } else s.skip_field((int)(t2 & 7));
            }
            if (!s.ok) return false;
        } else if (field == 3 && wire == 0) e.shard = (int)v.varint();
        else if (field == 4 && wire == 0) e.offset = (int64_t)v.varint();
        else if (field == 5 && wire == 0) e.size = (int64_t)v.varint();
        else if (field == 6 && wire == 5) e.crc32c = v.fixed32();
        else v.skip_field(wire);
    }
    return v.ok;
}

int ckpt_read_index(const char* index_path, CkptEntry* entries, int cap, int* n_out, char* err, size_t errcap) {
    std::vector<uint8_t> file;
    if (!read_file(index_path, file))
        return fail(err, errcap, ETHCNN_ERR_IO, std::string("cannot read ") + index_path);
    if (file.size() < 48) return fail(err, errcap, ETHCNN_ERR_FORMAT, "index shorter than its footer");
    const uint8_t* foot = file.data() + file.size() - 48;
    uint64_t magic;
    std::memcpy(&magic, foot + 40, 8);
    if (magic != 0xdb4775248b80fb57ull) return fail(err, errcap, ETHCNN_ERR_FORMAT, "bad table magic in .index");
    Cur fc{foot, foot + 40};
    uint64_t mo, ms, io, is;
    if (!parse_handle(fc, mo, ms) || !parse_handle(fc, io, is))
        return fail(err, errcap, ETHCNN_ERR_FORMAT, "bad footer handles");
    std::string why;
    Cur index_blk{nullptr, nullptr};
    if (!get_block(file, io, is, index_blk, why)) return fail(err, errcap, ETHCNN_ERR_FORMAT, why);
    int n = 0;
    bool header_seen = false, overflow = false;
    bool ok = for_each_entry(index_blk, [&](const std::string&, Cur hv) {
        uint64_t bo, bs;
        if (!parse_handle(hv, bo, bs)) return false;
        Cur data_blk{nullptr, nullptr};
        if (!get_block(file, bo, bs, data_blk, why)) return false;
        std::string why2;
        return for_each_entry(data_blk, [&](const std::string& key, Cur val) {
            if (key.empty()) { header_seen = true; return true; }  // BundleHeaderProto
            if (n >= cap) { overflow = true; return true; }
            CkptEntry& e = entries[n];
            if (key.size() >= sizeof(e.name)) return false;
            std::memset(e.name, 0, sizeof(e.name));
            std::memcpy(e.name, key.data(), key.size());
            if (!parse_entry_proto(val, e)) return false;
            ++n;
            return true;
        }, why2) || (why = why2, false);
    }, why);
    if (!ok) return fail(err, errcap, ETHCNN_ERR_FORMAT, std::string(index_path) + ": " + why);
    if (!header_seen) return fail(err, errcap, ETHCNN_ERR_FORMAT, "no bundle header entry in .index");
    if (overflow) return fail(err, errcap, ETHCNN_ERR_FORMAT, "more tensors in .index than expected");
    *n_out = n;
    return 0;
}

int ckpt_load_blob(const char* prefix, float* blob, char* err, size_t errcap) {
    return ckpt_load_table(prefix, kTensors, kNumTensors, blob, err, errcap);
}

int ckpt_load_table(const char* prefix, const TensorDesc* table, int ntensors, float* blob, char* err, size_t errcap) {
    CkptEntry ent[128];
    int n = 0;
    const std::string idx = std::string(prefix) + ".index";
    int rc = ckpt_read_index(idx.c_str(), ent, 128, &n, err, errcap);
    if (rc) return rc;
    std::vector<uint8_t> data;
    const std::string dpath = std::string(prefix) + ".data-00000-of-00001";
    if (!read_file(dpath, data)) return fail(err, errcap, ETHCNN_ERR_IO, "cannot read " + dpath);
    for (int t = 0; t < ntensors; ++t) {
        const TensorDesc& d = table[t];
        const CkptEntry* e = nullptr;
        for (int i = 0; i < n; ++i)
            if (std::strcmp(ent[i].name, d.name) == 0) e = &ent[i];
        if (!e) return fail(err, errcap, ETHCNN_ERR_FORMAT, std::string("checkpoint lacks tensor ") + d.name);
        bool shape_ok = (e->rank == d.rank) && e->dtype == 1 /* DT_FLOAT */ && e->shard == 0;
        for (int i = 0; shape_ok && i < d.rank; ++i) shape_ok = (e->shape[i] == d.shape[i]);
        if (!shape_ok || e->size != (int64_t)(d.count() * 4))
            return fail(err, errcap, ETHCNN_ERR_FORMAT, std::string("unexpected dtype/shape for ") + d.name);
        if (e->offset < 0 || (uint64_t)e->offset + (uint64_t)e->size > data.size())
            return fail(err, errcap, ETHCNN_ERR_FORMAT, std::string("tensor outside .data file: ") + d.name);
        const uint8_t* src = data.data() + e->offset;
        if (crc32c_mask(crc32c(src, (size_t)e->size)) != e->crc32c)
            return fail(err, errcap, ETHCNN_ERR_FORMAT, std::string("crc32c mismatch for ") + d.name);
        std::memcpy(blob + d.offset_bytes / 4, src, (size_t)e->size);
    }
    return 0;
}

}  // namespace ethcnn

extern "C" int ethcnn_ckpt_read_index(const char* index_path, ethcnn_ckpt_entry* entries, int cap, int* n_out,
                                      char* err, size_t errcap) {
    return ethcnn::ckpt_read_index(index_path, entries, cap, n_out, err, errcap);
}
extern "C" uint32_t ethcnn_crc32c_masked(const void* data, size_t n) {
    return ethcnn::crc32c_mask(ethcnn::crc32c(data, n));
}
extern "C" int ethcnn_ckpt_read_blob(const char* prefix, float* blob_out, size_t nfloats, char* err, size_t errcap) {
    if (!prefix || !blob_out || nfloats != ethcnn::kBlobFloats) return ETHCNN_ERR_ARG;
    return ethcnn::ckpt_load_blob(prefix, blob_out, err, errcap);
}
extern "C" int ethcnn_ckpt_read_lstm_blob(const char* prefix, float* blob_out, size_t nfloats, char* err, size_t errcap) {
    if (!prefix || !blob_out || nfloats != ethcnn::kLstmBlobFloats) return ETHCNN_ERR_ARG;
    return ethcnn::ckpt_load_table(prefix, ethcnn::kLstmTensors, ethcnn::kNumLstmTensors, blob_out, err, errcap);
}
