// fileformat_host.h -- granne's on-disk formats (host code, included by granne_hip.hip).
//
//   index file     /root/reference/src/index/io.rs:11-113: 1024-byte header "granne" + JSON, space
//                  padded; then per layer one MultiSetVector blob of layer_sizes[i] bytes
//   layer blob     src/slice_vector/offsets.rs:80-93,121-131 + set_vector.rs:164-222:
//                  [u64 LE bytes_for_offsets][Chunk x (1 + len/60)][data]
//                  Chunk (repr(C), 128 B) = { usize initial; u16 deltas[60] }, unused delta 0xFFFF;
//                  offset(j) = chunk[j/60].initial + sum_{t <= j%60} deltas[t]      (offsets.rs:177-187)
//   node record    set_vector.rs:91-148: [count u8] then either count raw LE u32 (when the record's
//                  payload is exactly 4*count bytes) or stream-vbyte (Scalar) of max(4,count)
//                  numbers; the numbers are deltas of the ascending neighbor ids
//   elements file  src/slice_vector/mod.rs:213-221, 460-466: [u64 LE width = dim][raw scalars]
//
// Reading decodes every node ONCE into a CSR pair that granne_hip_index_create_csr uploads; the
// GPU never sees the compressed form.
#pragma once
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

namespace granne_file {

constexpr size_t METADATA_LEN = 1024; // io.rs:7
constexpr size_t OFFSETS_PER_CHUNK = 60;
constexpr size_t CHUNK_BYTES = 128;

static inline uint64_t rd_u64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
static inline void wr_u64(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

// ---- the tiny part of JSON the header needs: "key": number | [numbers] --------------------------
static bool json_find(const std::string& js, const char* key, size_t* pos) {
    std::string k = std::string("\"") + key + "\"";
    size_t p = js.find(k);
    if (p == std::string::npos) return false;
    p = js.find(':', p + k.size());
    if (p == std::string::npos) return false;
    *pos = p + 1;
    return true;
}
// decimal digits at js[p..] -> *out; false on overflow of u64 (a header that long is corrupt)
static bool json_digits(const std::string& js, size_t* p, uint64_t* out) {
    uint64_t v = 0;
    while (*p < js.size() && isdigit((unsigned char)js[*p])) {
        const uint64_t d = (uint64_t)(js[*p] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
        ++*p;
    }
    *out = v;
    return true;
}
static bool json_number(const std::string& js, const char* key, uint64_t* out) {
    size_t p;
    if (!json_find(js, key, &p)) return false;
    while (p < js.size() && isspace((unsigned char)js[p])) ++p;
    if (p >= js.size() || !isdigit((unsigned char)js[p])) return false;
    return json_digits(js, &p, out);
}
static bool json_array(const std::string& js, const char* key, std::vector<uint64_t>* out) {
    size_t p;
    if (!json_find(js, key, &p)) return false;
    while (p < js.size() && isspace((unsigned char)js[p])) ++p;
    if (p >= js.size() || js[p] != '[') return false;
    ++p;
    out->clear();
    for (;;) {
        while (p < js.size() && (isspace((unsigned char)js[p]) || js[p] == ',')) ++p;
        if (p >= js.size()) return false;
        if (js[p] == ']') return true;
        if (!isdigit((unsigned char)js[p])) return false;
        uint64_t v = 0;
        if (!json_digits(js, &p, &v)) return false;
        out->push_back(v);
    }
}

// stream-vbyte 0.3.2, Scalar codec (Cargo.toml:42; call sites set_vector.rs:101,134)
static size_t svb_decode(const uint8_t* in, size_t in_len, size_t n, uint32_t* out) {
    size_t n_ctrl = (n + 3) / 4;
    if (n_ctrl > in_len) return (size_t)-1;
    size_t r = n_ctrl;
    for (size_t i = 0; i < n; ++i) {
        unsigned len = ((in[i / 4] >> (2 * (i % 4))) & 3u) + 1;
        if (r + len > in_len) return (size_t)-1;
        uint32_t v = 0;
        for (unsigned k = 0; k < len; ++k) v |= (uint32_t)in[r++] << (8 * k);
        out[i] = v;
    }
    return r;
}
static size_t svb_encode(const uint32_t* in, size_t n, uint8_t* out) {
    size_t n_ctrl = (n + 3) / 4;
    memset(out, 0, n_ctrl);
    size_t w = n_ctrl;
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = in[i];
        unsigned len = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
        out[i / 4] |= (uint8_t)((len - 1) << (2 * (i % 4)));
        for (unsigned k = 0; k < len; ++k) out[w++] = (uint8_t)(v >> (8 * k));
    }
    return w;
}

// decode_into, set_vector.rs:91-115. Returns count or -1.
static int decode_node(const uint8_t* rec, size_t len, uint32_t* out /* >= 256 */) {
    if (len < 1) return -1;
    size_t count = rec[0];
    const uint8_t* p = rec + 1;
    size_t plen = len - 1;
    if (plen != count * 4) {
        uint32_t tmp[256];
        size_t m = count < 4 ? 4 : count; // MIN_NUMBERS_TO_ENCODE
        if (svb_decode(p, plen, m, tmp) == (size_t)-1) return -1;
        for (size_t i = 0; i < count; ++i) out[i] = tmp[i];
    } else {
        for (size_t i = 0; i < count; ++i)
            out[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 |
                     (uint32_t)p[4 * i + 3] << 24;
    }
    for (size_t i = 1; i < count; ++i) out[i] += out[i - 1]; // delta_decode, :157-162
    return (int)count;
}

// set_encode, set_vector.rs:117-148 (ids must be ascending). out >= 1 + 5*256.
static size_t encode_node(const uint32_t* sorted, size_t n, uint8_t* out) {
    if (n > 255) n = 255;
    uint32_t buf[256];
    for (size_t i = 0; i < n; ++i) buf[i] = sorted[i];
    for (size_t i = n; i > 1; --i) buf[i - 1] -= buf[i - 2]; // delta_encode, :150-155
    size_t count = n;
    size_t m = n < 4 ? 4 : n;
    for (size_t i = n; i < m; ++i) buf[i] = 0;
    size_t enc = svb_encode(buf, m, out + 1);
    if (enc >= 4 * count) { // keep the raw form unless compression makes it smaller, :137-143
        for (size_t i = 0; i < count; ++i)
            for (int b = 0; b < 4; ++b) out[1 + 4 * i + b] = (uint8_t)(buf[i] >> (8 * b));
        enc = 4 * count;
    }
    out[0] = (uint8_t)count;
    return enc + 1;
}

struct DecodedLayer {
    std::vector<uint64_t> offsets; // len + 1
    std::vector<uint32_t> ids;
};

// one MultiSetVector blob -> CSR
static int decode_layer(const uint8_t* blob, size_t size, uint64_t expect_len, DecodedLayer* L, std::string* err) {
    if (size < 8) { *err = "layer blob too small"; return -1; }
    uint64_t off_bytes = rd_u64(blob);
    if (off_bytes % CHUNK_BYTES != 0 || off_bytes > size - 8) { *err = "bad offsets size in layer blob"; return -1; }
    const uint8_t* chunks = blob + 8;
    size_t n_chunks = off_bytes / CHUNK_BYTES;
    const uint8_t* data = chunks + off_bytes;
    size_t data_len = size - 8 - off_bytes;
    // number of offsets = 60 * (n_chunks - 1) + used deltas of the last chunk (offsets.rs:270-272)
    if (n_chunks == 0) { *err = "layer blob without offset chunks"; return -1; }
    const uint8_t* last = chunks + (n_chunks - 1) * CHUNK_BYTES;
    size_t used = 0;
    while (used < OFFSETS_PER_CHUNK && ((uint16_t)last[8 + 2 * used] | (uint16_t)last[9 + 2 * used] << 8) != 0xFFFF) ++used;
    uint64_t n_off = OFFSETS_PER_CHUNK * (n_chunks - 1) + used;
    if (n_off == 0) { *err = "layer blob without offsets"; return -1; }
    uint64_t len = n_off - 1;
    if (len != expect_len) { *err = "layer_counts disagrees with the layer's offset table"; return -1; }
    L->offsets.assign(len + 1, 0);
    L->ids.clear();
    L->ids.reserve(len * 16);
    uint64_t prev = 0;
    uint32_t tmp[256];
    for (uint64_t j = 0; j <= len; ++j) {
        const uint8_t* c = chunks + (j / OFFSETS_PER_CHUNK) * CHUNK_BYTES;
        uint64_t o = rd_u64(c);
        for (size_t t = 0; t <= j % OFFSETS_PER_CHUNK; ++t) o += (uint16_t)c[8 + 2 * t] | (uint16_t)c[9 + 2 * t] << 8;
        if (o < prev || o > data_len) { *err = "offsets are not monotone / exceed the data"; return -1; }
        if (j > 0) {
            int cnt = decode_node(data + prev, (size_t)(o - prev), tmp);
            if (cnt < 0) { *err = "malformed neighbor record"; return -1; }
            L->ids.insert(L->ids.end(), tmp, tmp + cnt);
            L->offsets[j] = L->ids.size();
        }
        prev = o;
    }
    return 0;
}

// the whole index file -> per-layer CSR
static int decode_index(const uint8_t* buf, size_t len, std::vector<DecodedLayer>* layers, std::string* err) {
    if (len < METADATA_LEN || memcmp(buf, "granne", 6) != 0) { *err = "Library string missing"; return -1; } // io.rs:99-101
    std::string js((const char*)buf + 6, METADATA_LEN - 6);
    uint64_t num_layers = 0;
    std::vector<uint64_t> counts, sizes;
    if (!json_number(js, "num_layers", &num_layers) || !json_array(js, "layer_counts", &counts) ||
        !json_array(js, "layer_sizes", &sizes)) { *err = "Could not read metadata"; return -1; }
    if (counts.size() != num_layers || sizes.size() != num_layers) { *err = "metadata arrays disagree with num_layers"; return -1; }
    size_t start = METADATA_LEN;
    layers->resize(num_layers);
    for (uint64_t l = 0; l < num_layers; ++l) {
        if (sizes[l] > len - start) { *err = "index file truncated"; return -1; } // no wrap: start <= len
        if (decode_layer(buf + start, sizes[l], counts[l], &(*layers)[l], err)) return -1;
        start += sizes[l];
    }
    return 0;
}

// write_index (io.rs:11-70) from fixed-width rows
static int encode_index(uint32_t n_layers, const uint64_t* layer_len, const uint32_t* const* rows, const uint32_t* width,
                        std::vector<uint8_t>* out) {
    out->assign(METADATA_LEN, (uint8_t)' ');
    std::vector<uint64_t> sizes;
    uint64_t num_neighbors = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        const uint64_t len = layer_len[l];
        const size_t n_chunks = 1 + len / OFFSETS_PER_CHUNK; // set_vector.rs:176-177
        const size_t blob0 = out->size();
        out->resize(blob0 + 8 + n_chunks * CHUNK_BYTES);
        wr_u64(out->data() + blob0, n_chunks * CHUNK_BYTES);
        // the nodes' records, encoded by up to 16 host threads over consecutive node ranges (one thread took 69 s for a
        // 125M-node layer), laid end to end in node order; offs[i + 1] = bytes of records 0..i
        std::vector<uint64_t> offs(len + 1, 0);
        unsigned T = 1;
        if (len >= (1u << 16)) {
            T = std::thread::hardware_concurrency();
            T = T < 1 ? 1 : (T > 16 ? 16 : T);
        }
        std::vector<std::vector<uint8_t>> part(T);
        uint64_t first_count = 0;
        auto work = [&](unsigned t) {
            const uint64_t i0 = len * t / T, i1 = len * (t + 1) / T;
            std::vector<uint32_t> mine;
            uint8_t r_[1 + 5 * 256];
            std::vector<uint8_t>& dst = part[t];
            dst.reserve((size_t)(i1 - i0) * (width[l] < 8 ? 8 : width[l]) * 2);
            for (uint64_t i = i0; i < i1; ++i) {
                const uint32_t* r = rows[l] + i * width[l];
                mine.clear();
                for (uint32_t k = 0; k < width[l]; ++k)
                    if (r[k] != GRANNE_HIP_UNUSED) mine.push_back(r[k]); // predicate x != UNUSED, io.rs:31
                if (i == 0) first_count = mine.size();
                std::sort(mine.begin(), mine.end());
                const size_t n = encode_node(mine.data(), mine.size(), r_);
                dst.insert(dst.end(), r_, r_ + n);
                offs[i + 1] = n;
            }
        };
        if (T == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        if (l + 1 == n_layers && len) num_neighbors = first_count; // get_neighbors(0).len(), io.rs:20-24
        for (uint64_t i = 0; i < len; ++i) offs[i + 1] += offs[i];
        for (unsigned t = 0; t < T; ++t) {
            out->insert(out->end(), part[t].begin(), part[t].end());
            std::vector<uint8_t>().swap(part[t]);
        }
        // offsets: chunks of 60, each starting with initial = its first offset and delta 0
        uint8_t* ch = out->data() + blob0 + 8;
        for (size_t c = 0; c < n_chunks; ++c) {
            uint8_t* p = ch + c * CHUNK_BYTES;
            size_t first = c * OFFSETS_PER_CHUNK;
            uint64_t initial = first <= len ? offs[first] : 0;
            wr_u64(p, initial);
            uint64_t prev = initial;
            for (size_t t = 0; t < OFFSETS_PER_CHUNK; ++t) {
                uint16_t d = 0xFFFF;
                if (first + t <= len) {
                    uint64_t delta = offs[first + t] - prev;
                    if (delta >= 0xFFFF) return -1;
                    d = (uint16_t)delta;
                    prev = offs[first + t];
                }
                p[8 + 2 * t] = (uint8_t)d;
                p[9 + 2 * t] = (uint8_t)(d >> 8);
            }
        }
        sizes.push_back(out->size() - blob0);
    }
    // metadata: serde_json's default map is sorted by key (io.rs:46-63)
    std::string js = "granne{\"compressed\":true,\"granne_version\":\"0.5.2\",\"layer_counts\":[";
    for (uint32_t l = 0; l < n_layers; ++l) js += (l ? "," : "") + std::to_string(layer_len[l]);
    js += "],\"layer_sizes\":[";
    for (uint32_t l = 0; l < n_layers; ++l) js += (l ? "," : "") + std::to_string(sizes[l]);
    js += "],\"num_elements\":" + std::to_string(n_layers ? layer_len[n_layers - 1] : 0);
    js += ",\"num_layers\":" + std::to_string(n_layers);
    js += ",\"num_neighbors\":" + std::to_string(num_neighbors) + ",\"version\":2}";
    if (js.size() > METADATA_LEN) return -1;
    memcpy(out->data(), js.data(), js.size());
    return 0;
}

struct MappedFile {
    const uint8_t* data = nullptr;
    size_t len = 0;
    int fd = -1;
    bool open_ro(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        len = (size_t)st.st_size;
        if (len == 0) { data = (const uint8_t*)""; return true; }
        void* p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) return false;
        data = (const uint8_t*)p;
        return true;
    }
    ~MappedFile() {
        if (data && len) munmap((void*)data, len);
        if (fd >= 0) ::close(fd);
    }
};

static bool write_file(const char* path, const void* a, size_t alen, const void* b, size_t blen) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    bool ok = (alen == 0 || fwrite(a, 1, alen, f) == alen) && (blen == 0 || fwrite(b, 1, blen, f) == blen);
    return fclose(f) == 0 && ok;
}

} // namespace granne_file

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int granne_hip_index_load(granne_hip_index** out, const void* index_bytes, uint64_t index_len,
                                     const void* elements_bytes, uint64_t elements_len, int dtype, int device_id) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!index_bytes || !elements_bytes) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (dtype != GRANNE_HIP_F32 && dtype != GRANNE_HIP_I8) return fail(GRANNE_HIP_ERR_INVALID, "unknown dtype %d", dtype);
    // elements: [u64 width][scalars] (src/slice_vector/mod.rs:213-221); width > 0, len % width == 0 (:116-118)
    if (elements_len < 8) return fail(GRANNE_HIP_ERR_IO, "elements file too small");
    const uint8_t* eb = (const uint8_t*)elements_bytes;
    uint64_t dim = granne_file::rd_u64(eb);
    uint64_t esz = elem_size(dtype);
    uint64_t payload = elements_len - 8;
    if (dim == 0 || dim > 0xFFFFFFFFull || payload % esz != 0 || (payload / esz) % dim != 0)
        return fail(GRANNE_HIP_ERR_IO, "elements file: width %llu does not divide the data", (unsigned long long)dim);
    uint64_t n = payload / esz / dim;
    std::vector<granne_file::DecodedLayer> layers;
    std::string err;
    if (granne_file::decode_index((const uint8_t*)index_bytes, index_len, &layers, &err))
        return fail(GRANNE_HIP_ERR_IO, "index file: %s", err.c_str());
    std::vector<uint64_t> lens(layers.size());
    std::vector<const uint64_t*> offs(layers.size());
    std::vector<const uint32_t*> ids(layers.size());
    for (size_t l = 0; l < layers.size(); ++l) {
        lens[l] = layers[l].offsets.size() - 1;
        offs[l] = layers[l].offsets.data();
        ids[l] = layers[l].ids.data();
        for (uint32_t id : layers[l].ids)
            if (id >= lens[l]) return fail(GRANNE_HIP_ERR_IO, "index file: neighbor id outside its layer");
    }
    return granne_hip_index_create_csr(out, eb + 8, n, (uint32_t)dim, dtype, (uint32_t)layers.size(), lens.data(),
                                       offs.data(), ids.data(), device_id);
}

extern "C" int granne_hip_index_load_files(granne_hip_index** out, const char* index_path, const char* elements_path,
                                           int dtype, int device_id) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!index_path || !elements_path) return fail(GRANNE_HIP_ERR_INVALID, "null path");
    granne_file::MappedFile fi, fe;
    if (!fi.open_ro(index_path)) return fail(GRANNE_HIP_ERR_IO, "Could not open index file %s", index_path);
    if (!fe.open_ro(elements_path)) return fail(GRANNE_HIP_ERR_IO, "Could not open elements file %s", elements_path);
    return granne_hip_index_load(out, fi.data, fi.len, fe.data, fe.len, dtype, device_id);
}

extern "C" int granne_hip_write_index_file(const char* path, uint32_t n_layers, const uint64_t* layer_len,
                                           const uint32_t* const* layer_rows, const uint32_t* layer_width) {
    if (!path || (n_layers && (!layer_len || !layer_rows || !layer_width))) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    std::vector<uint8_t> buf;
    if (granne_file::encode_index(n_layers, layer_len, layer_rows, layer_width, &buf))
        return fail(GRANNE_HIP_ERR_IO, "index does not fit the file format");
    if (!granne_file::write_file(path, buf.data(), buf.size(), nullptr, 0)) return fail(GRANNE_HIP_ERR_IO, "Could not write %s", path);
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_write_elements_file(const char* path, const void* elements, uint64_t n_elements, uint32_t dim,
                                              int dtype) {
    if (!path || (n_elements && !elements)) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (dtype != GRANNE_HIP_F32 && dtype != GRANNE_HIP_I8) return fail(GRANNE_HIP_ERR_INVALID, "unknown dtype %d", dtype);
    uint8_t hdr[8];
    granne_file::wr_u64(hdr, dim);
    if (!granne_file::write_file(path, hdr, 8, elements, (size_t)n_elements * dim * elem_size(dtype)))
        return fail(GRANNE_HIP_ERR_IO, "Could not write %s", path);
    return GRANNE_HIP_OK;
}

// the index file's bytes of a device-resident index (downloads the layers, encodes them)
static int encode_device_index(const granne_hip_index* ix, std::vector<uint8_t>* buf) {
    std::vector<std::vector<uint32_t>> rows(ix->layers.size());
    std::vector<uint64_t> lens(ix->layers.size());
    std::vector<const uint32_t*> ptrs(ix->layers.size());
    std::vector<uint32_t> widths(ix->layers.size());
    for (size_t l = 0; l < ix->layers.size(); ++l) {
        const LayerHost& L = ix->layers[l];
        rows[l].resize((size_t)L.len * L.dev_width);
        if (!rows[l].empty())
            HIP_TRY(hipMemcpy(rows[l].data(), L.d_adj, rows[l].size() * 4, hipMemcpyDeviceToHost));
        lens[l] = L.len;
        ptrs[l] = rows[l].data();
        widths[l] = L.dev_width;
    }
    if (granne_file::encode_index((uint32_t)rows.size(), lens.data(), ptrs.data(), widths.data(), buf))
        return fail(GRANNE_HIP_ERR_IO, "index does not fit the file format");
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_encode(const granne_hip_index* ix, void** out_bytes, uint64_t* out_len) {
    if (!ix || !out_bytes || !out_len) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    *out_bytes = nullptr;
    *out_len = 0;
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    std::vector<uint8_t> buf;
    int rc = encode_device_index(ix, &buf);
    if (rc) return rc;
    void* p = malloc(buf.size() ? buf.size() : 1);
    if (!p) return fail(GRANNE_HIP_ERR_IO, "out of host memory (%zu bytes)", buf.size());
    memcpy(p, buf.data(), buf.size());
    *out_bytes = p;
    *out_len = buf.size();
    return GRANNE_HIP_OK;
}
extern "C" void granne_hip_bytes_free(void* bytes) { free(bytes); }

// Index::write_index / write_elements for a device-resident index (downloads, then writes)
extern "C" int granne_hip_index_save(const granne_hip_index* ix, const char* index_path, const char* elements_path) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    if (index_path) {
        std::vector<uint8_t> buf;
        int rc = encode_device_index(ix, &buf);
        if (rc) return rc;
        if (!granne_file::write_file(index_path, buf.data(), buf.size(), nullptr, 0)) return fail(GRANNE_HIP_ERR_IO, "Could not write %s", index_path);
    }
    if (elements_path) {
        size_t dense = (size_t)ix->dim * elem_size(ix->dtype);
        std::vector<uint8_t> el((size_t)ix->n_elements * dense);
        if (!el.empty())
            HIP_TRY(hipMemcpy2D(el.data(), dense, ix->d_elements, ix->row_stride, dense, ix->n_elements, hipMemcpyDeviceToHost));
        return granne_hip_write_elements_file(elements_path, el.data(), ix->n_elements, ix->dim, ix->dtype);
    }
    return GRANNE_HIP_OK;
}

// host-only inspection of an index file (no device needed)
extern "C" int granne_hip_index_file_info(const void* index_bytes, uint64_t index_len, uint32_t* out_n_layers,
                                          uint64_t* out_layer_len, uint64_t* out_layer_ids, uint32_t cap) {
    if (!index_bytes || !out_n_layers) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    std::vector<granne_file::DecodedLayer> layers;
    std::string err;
    if (granne_file::decode_index((const uint8_t*)index_bytes, index_len, &layers, &err))
        return fail(GRANNE_HIP_ERR_IO, "index file: %s", err.c_str());
    *out_n_layers = (uint32_t)layers.size();
    for (size_t l = 0; l < layers.size() && l < cap; ++l) {
        if (out_layer_len) out_layer_len[l] = layers[l].offsets.size() - 1;
        if (out_layer_ids) out_layer_ids[l] = layers[l].ids.size();
    }
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_file_decode_layer(const void* index_bytes, uint64_t index_len, uint32_t layer,
                                                  uint64_t* out_offsets, uint32_t* out_ids) {
    if (!index_bytes || !out_offsets) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    std::vector<granne_file::DecodedLayer> layers;
    std::string err;
    if (granne_file::decode_index((const uint8_t*)index_bytes, index_len, &layers, &err))
        return fail(GRANNE_HIP_ERR_IO, "index file: %s", err.c_str());
    if (layer >= layers.size()) return fail(GRANNE_HIP_ERR_INVALID, "layer out of range");
    memcpy(out_offsets, layers[layer].offsets.data(), layers[layer].offsets.size() * 8);
    if (out_ids && !layers[layer].ids.empty()) memcpy(out_ids, layers[layer].ids.data(), layers[layer].ids.size() * 4);
    return GRANNE_HIP_OK;
}

// GranneBuilder::from_bytes (src/index/mod.rs:430-461): a fresh builder adopts the layers of a written
// index; every neighbor list is resized to config.num_neighbors (:448 -- truncated, or padded with UNUSED).
extern "C" int granne_hip_builder_load_index(granne_hip_builder* b, const void* index_bytes, uint64_t index_len) {
    if (!b) return fail(GRANNE_HIP_ERR_INVALID, "builder is null");
    if (!index_bytes) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (!b->layers.empty()) return fail(GRANNE_HIP_ERR_INVALID, "the builder already has layers");
    std::vector<granne_file::DecodedLayer> layers;
    std::string err;
    if (granne_file::decode_index((const uint8_t*)index_bytes, index_len, &layers, &err))
        return fail(GRANNE_HIP_ERR_IO, "index file: %s", err.c_str());
    uint64_t prev = 0;
    for (size_t l = 0; l < layers.size(); ++l) {
        const uint64_t len = layers[l].offsets.size() - 1;
        if (len < prev) return fail(GRANNE_HIP_ERR_IO, "index file: layers are not prefix-nested");
        if (len > b->n_elements) return fail(GRANNE_HIP_ERR_IO, "index file: more nodes than the builder has elements");
        for (uint32_t id : layers[l].ids)
            if (id >= len) return fail(GRANNE_HIP_ERR_IO, "index file: neighbor id outside its layer");
        prev = len;
    }
    if (layers.size() > 64) return fail(GRANNE_HIP_ERR_INVALID, "too many layers");
    DeviceGuard g(b->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", b->device);
    const uint32_t nn = b->cfg.num_neighbors, W = b->W;
    std::vector<BuilderLayer> fresh;
    auto drop = [&]() {
        for (auto& L : fresh)
            if (L.d_adj) (void)hipFree(L.d_adj);
    };
    for (size_t l = 0; l < layers.size(); ++l) {
        const granne_file::DecodedLayer& D = layers[l];
        const uint64_t len = D.offsets.size() - 1;
        std::vector<uint32_t> rows((size_t)len * W, 0xFFFFFFFFu);
        for (uint64_t i = 0; i < len; ++i) {
            const uint64_t cnt = D.offsets[i + 1] - D.offsets[i];
            const uint64_t keep = cnt < nn ? cnt : nn; // neighbors.resize(num_neighbors, UNUSED)
            for (uint64_t c = 0; c < keep; ++c) rows[(size_t)i * W + c] = D.ids[D.offsets[i] + c];
        }
        BuilderLayer L;
        L.len = len;
        L.cap_rows = len;
        size_t bytes = rows.size() * 4;
        hipError_t e = hipMalloc((void**)&L.d_adj, bytes ? bytes : 16);
        if (e == hipSuccess && bytes) e = hipMemcpy(L.d_adj, rows.data(), bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (L.d_adj) (void)hipFree(L.d_adj);
            drop();
            return fail(GRANNE_HIP_ERR_HIP, "layer upload failed: %s", hipGetErrorString(e));
        }
        b->hbm_bytes += bytes;
        fresh.push_back(L);
    }
    b->layers = fresh;
    return GRANNE_HIP_OK;
}

