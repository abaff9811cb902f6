// granne.hpp -- C++17 host-side mirror of granne's Rust API for the search path, over the C ABI of
// include/granne_hip.h. The reference is Rust (no toolchain in this environment); this header
// keeps its names, argument meaning and error behaviour so that code -- and tests -- read like the
// reference's:
//
//   granne::angular::Vector / Vectors          src/elements/angular.rs, dense_vector.rs
//   granne::angular_int::Vector / Vectors      src/elements/angular_int.rs
//   granne::BuildConfig                        src/index/mod.rs:198-291 (fluent setters)
//   granne::GranneBuilder<Elements>            src/index/mod.rs:293-531   (build, build_partial, get_index, ...)
//   granne::Granne<Elements>                   src/index/mod.rs:38-185    (search, len, num_layers, ...)
//
// Where the reference panics (max_search == 0, malformed files) these throw std::runtime_error
// carrying granne_hip_last_error(). Vectors own host copies of their rows (like the Rust
// `Vectors<'static>`); a Granne/GranneBuilder owns the device copies.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "granne_hip.h"

namespace granne {

inline void check(int rc) {
    if (rc != GRANNE_HIP_OK) throw std::runtime_error(std::string("granne_hip: ") + granne_hip_last_error());
}

namespace detail {
template <class Scalar> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = GRANNE_HIP_F32; };
template <> struct dtype_of<int8_t> { static constexpr int value = GRANNE_HIP_I8; };

// dense_vector! (src/elements/dense_vector.rs): Vector = one row, Vectors = row-major collection
template <class Scalar>
struct Vector {
    std::vector<Scalar> data;
    size_t len() const { return data.size(); }
    const Scalar* as_slice() const { return data.data(); }
};

template <class Scalar>
class Vectors {
public:
    using Element = Vector<Scalar>;
    Vectors() = default;
    // Vectors::from_vec(vec, dim): rows are taken as stored (already normalised / quantised)
    static Vectors from_vec(std::vector<Scalar> v, size_t dim) {
        if (dim == 0 || v.size() % dim != 0) throw std::runtime_error("dim must be non-zero and divide the length");
        Vectors r;
        r.dim_ = dim;
        r.data_ = std::move(v);
        return r;
    }
    void push(const Element& e) { // the dimension is set by the first vector pushed
        if (dim_ == 0) dim_ = e.len();
        if (e.len() != dim_) throw std::runtime_error("dimension mismatch");
        data_.insert(data_.end(), e.data.begin(), e.data.end());
    }
    size_t len() const { return dim_ ? data_.size() / dim_ : 0; }
    size_t dim() const { return dim_; }
    Element get_element(size_t i) const { return Element{std::vector<Scalar>(data_.begin() + i * dim_, data_.begin() + (i + 1) * dim_)}; }
    const Scalar* as_slice() const { return data_.data(); }

private:
    size_t dim_ = 0;
    std::vector<Scalar> data_;
};
} // namespace detail

namespace angular {
using Vector = detail::Vector<float>;
using Vectors = detail::Vectors<float>;
// impl From<Vec<f32>> for Vector (angular.rs:55-61): normalised on construction, on the device
inline Vector from(std::vector<float> v, int device = 0) {
    if (!v.empty()) check(granne_hip_normalize_f32(v.data(), 1, (uint32_t)v.size(), device));
    return Vector{std::move(v)};
}
} // namespace angular

namespace angular_int {
using Vector = detail::Vector<int8_t>;
using Vectors = detail::Vectors<int8_t>;
// Vector::quantize (angular_int.rs:27-45)
inline Vector from(const std::vector<float>& v, int device = 0) {
    Vector r{std::vector<int8_t>(v.size())};
    if (!v.empty()) check(granne_hip_quantize_f32(v.data(), r.data.data(), 1, (uint32_t)v.size(), device));
    return r;
}
} // namespace angular_int

// BuildConfig (src/index/mod.rs:198-291)
class BuildConfig {
public:
    BuildConfig() { granne_hip_build_config_default(&c_); }
    static BuildConfig new_() { return BuildConfig(); }
    BuildConfig num_neighbors(size_t n) const { BuildConfig r = *this; r.c_.num_neighbors = (uint32_t)n; return r; }
    BuildConfig max_search(size_t n) const { BuildConfig r = *this; r.c_.max_search = (uint32_t)n; return r; }
    BuildConfig expected_num_elements(size_t n) const { BuildConfig r = *this; r.c_.expected_num_elements = n; return r; }
    BuildConfig layer_multiplier(float m) const { BuildConfig r = *this; r.c_.layer_multiplier = m; return r; }
    BuildConfig reinsert_elements(bool yes) const { BuildConfig r = *this; r.c_.reinsert_elements = yes; return r; }
    BuildConfig show_progress(bool yes) const { BuildConfig r = *this; r.c_.show_progress = yes; return r; }
    const granne_hip_build_config& raw() const { return c_; }

private:
    granne_hip_build_config c_;
};

// Granne (src/index/mod.rs:38-185): search + the Index trait
template <class Elements>
class Granne {
public:
    using Element = typename Elements::Element;
    explicit Granne(granne_hip_index* h) : h_(h, granne_hip_index_destroy) {}

    // Granne::from_file (mod.rs:122-135) over Vectors::from_file
    static Granne from_file(const std::string& index_path, const std::string& elements_path, int device = 0) {
        granne_hip_index* h = nullptr;
        check(granne_hip_index_load_files(&h, index_path.c_str(), elements_path.c_str(), dtype(), device));
        return Granne(h);
    }
    // Granne::from_bytes (mod.rs:106-113)
    static Granne from_bytes(const void* index, size_t index_len, const void* elements, size_t elements_len, int device = 0) {
        granne_hip_index* h = nullptr;
        check(granne_hip_index_load(&h, index, index_len, elements, elements_len, dtype(), device));
        return Granne(h);
    }

    // Granne::search(&element, max_search, num_neighbors) -> Vec<(usize, f32)> (mod.rs:140-150)
    std::vector<std::pair<size_t, float>> search(const Element& element, size_t max_search, size_t num_neighbors) const {
        return std::move(search_batch(&element, 1, max_search, num_neighbors)[0]);
    }
    // nq independent searches in one launch
    std::vector<std::vector<std::pair<size_t, float>>> search_batch(const Element* elements, size_t nq, size_t max_search,
                                                                    size_t num_neighbors) const {
        const size_t dim = granne_hip_index_dim(h_.get());
        std::vector<typename decltype(Element::data)::value_type> q(nq * dim);
        for (size_t i = 0; i < nq; ++i) {
            if (elements[i].len() != dim) throw std::runtime_error("query dimension mismatch");
            std::memcpy(q.data() + i * dim, elements[i].as_slice(), dim * sizeof(q[0]));
        }
        std::vector<uint64_t> ids(nq * num_neighbors);
        std::vector<float> ds(nq * num_neighbors);
        std::vector<uint32_t> counts(nq);
        check(granne_hip_search_batch(h_.get(), q.data(), (uint32_t)nq, (uint32_t)max_search, (uint32_t)num_neighbors,
                                      ids.data(), ds.data(), counts.data(), nullptr));
        std::vector<std::vector<std::pair<size_t, float>>> out(nq);
        for (size_t i = 0; i < nq; ++i)
            for (uint32_t j = 0; j < counts[i]; ++j) out[i].emplace_back((size_t)ids[i * num_neighbors + j], ds[i * num_neighbors + j]);
        return out;
    }

    // The exact k nearest elements of every query by a scan of all elements on the matrix cores (k <= 16): what
    // ElementContainer::dists over every index + a sort would give (src/elements/mod.rs:35-39); the recall ground truth.
    std::vector<std::vector<std::pair<size_t, float>>> brute_force(const Element* elements, size_t nq, size_t k) const {
        const size_t dim = granne_hip_index_dim(h_.get());
        std::vector<typename decltype(Element::data)::value_type> q(nq * dim);
        for (size_t i = 0; i < nq; ++i) {
            if (elements[i].len() != dim) throw std::runtime_error("query dimension mismatch");
            std::memcpy(q.data() + i * dim, elements[i].as_slice(), dim * sizeof(q[0]));
        }
        std::vector<uint64_t> ids(nq * k);
        std::vector<float> ds(nq * k);
        std::vector<uint32_t> counts(nq);
        check(granne_hip_brute_force(h_.get(), q.data(), (uint32_t)nq, (uint32_t)k, ids.data(), ds.data(), counts.data()));
        std::vector<std::vector<std::pair<size_t, float>>> out(nq);
        for (size_t i = 0; i < nq; ++i)
            for (uint32_t j = 0; j < counts[i]; ++j) out[i].emplace_back((size_t)ids[i * k + j], ds[i * k + j]);
        return out;
    }

    // Index trait (mod.rs:54-71)
    size_t len() const { return granne_hip_index_len(h_.get()); }
    size_t num_layers() const { return granne_hip_index_num_layers(h_.get()); }
    size_t layer_len(size_t layer) const { return granne_hip_index_layer_len(h_.get(), (uint32_t)layer); }
    std::vector<size_t> get_neighbors(size_t index, size_t layer) const {
        uint32_t buf[512], n = 0;
        check(granne_hip_index_get_neighbors(h_.get(), index, (uint32_t)layer, buf, 512, &n));
        return std::vector<size_t>(buf, buf + n);
    }
    Element get_element(size_t index) const {
        Element e;
        e.data.resize(granne_hip_index_dim(h_.get()));
        check(granne_hip_index_get_element(h_.get(), index, e.data.data()));
        return e;
    }
    // Granne::reorder / reorder_by_keys (src/index/reorder.rs:59-133): returns the permutation
    std::vector<size_t> reorder(bool /*show_progress*/ = false) {
        std::vector<uint64_t> order(len());
        check(granne_hip_index_reorder(h_.get(), order.data()));
        return std::vector<size_t>(order.begin(), order.end());
    }
    std::vector<size_t> reorder_by_keys(const std::vector<uint64_t>& keys, bool /*show_progress*/ = false) {
        if (keys.size() != len()) throw std::runtime_error("need one key per element"); // reorder.rs:91
        std::vector<uint64_t> order(len());
        check(granne_hip_index_reorder_by_keys(h_.get(), keys.data(), order.data()));
        return std::vector<size_t>(order.begin(), order.end());
    }
    void write_index(const std::string& path) const { check(granne_hip_index_save(h_.get(), path.c_str(), nullptr)); }
    void write_elements(const std::string& path) const { check(granne_hip_index_save(h_.get(), nullptr, path.c_str())); }
    // Index::write_index into a writer (src/index/mod.rs:67-70): the bytes of the index file
    std::vector<uint8_t> index_bytes() const {
        void* p = nullptr;
        uint64_t n = 0;
        check(granne_hip_index_encode(h_.get(), &p, &n));
        std::vector<uint8_t> out(static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n);
        granne_hip_bytes_free(p);
        return out;
    }
    // the walkers' copy of the layers with the neighbors' row tails next to the ids (f32 100-d / 200-d: whole-line reads)
    void set_inline_tails(bool keep) { check(granne_hip_index_set_option(h_.get(), GRANNE_HIP_OPT_INLINE_TAILS, keep ? 1 : 0)); }
    // from how many walks per launch revisits are skipped before their rows are fetched (f32; 0 = always)
    void set_seen_min(uint64_t walks) { check(granne_hip_index_set_option(h_.get(), GRANNE_HIP_OPT_SEEN_MIN, walks)); }
    granne_hip_index* raw() const { return h_.get(); }

private:
    static constexpr int dtype() { return detail::dtype_of<typename decltype(Element::data)::value_type>::value; }
    std::shared_ptr<granne_hip_index> h_;
};

// A partitioned index: shard s is a Granne of its own over the elements [offsets[s], offsets[s] + shard.len()) of the
// whole set -- how the reference's shard helper cuts an element file (src/elements/embeddings/parsing.rs:63-100). A search
// asks every shard and keeps the best by (distance, global id); ids in the results are global. One host process drives
// all shards (granne_hip_sharded_*); batches are pipelined inside the library.
template <class Elements>
class ShardedGranne {
public:
    using Element = typename Elements::Element;
    ShardedGranne(std::vector<Granne<Elements>> shards, const std::vector<uint64_t>& offsets) : shards_(std::move(shards)) {
        if (shards_.empty() || shards_.size() != offsets.size()) throw std::runtime_error("need one offset per shard");
        std::vector<granne_hip_index*> hs;
        for (auto& s : shards_) hs.push_back(s.raw());
        granne_hip_sharded* h = nullptr;
        check(granne_hip_sharded_create(&h, hs.data(), offsets.data(), (uint32_t)hs.size()));
        h_.reset(h, granne_hip_sharded_destroy); // destroyed before shards_ (declared after it): the handle borrows them
    }
    // the whole element set in, a searchable partitioned index out (granne_hip_sharded_build): n_shards id ranges
    // (src/elements/embeddings/parsing.rs:72-98), each built with the GPU builder under `config`; the handle owns its shards
    static ShardedGranne build(const BuildConfig& config, const Elements& elements, size_t n_shards, const std::vector<int>& devices = {0}) {
        using Scalar = typename decltype(Elements::Element::data)::value_type;
        granne_hip_sharded* h = nullptr;
        check(granne_hip_sharded_build(&h, &config.raw(), elements.as_slice(), elements.len(), (uint32_t)(elements.dim() ? elements.dim() : 1),
                                       detail::dtype_of<Scalar>::value, (uint32_t)n_shards, devices.data(), (uint32_t)devices.size()));
        ShardedGranne s;
        s.h_.reset(h, granne_hip_sharded_destroy);
        return s;
    }
    size_t len() const { return granne_hip_sharded_len(h_.get()); }
    size_t num_shards() const { return granne_hip_sharded_num_shards(h_.get()); }
    // the exchange step as ONE RCCL all-gather over the shard devices instead of peer copies (librccl by dlopen)
    void use_rccl_all_gather() { check(granne_hip_sharded_set_option(h_.get(), GRANNE_HIP_SHARDED_OPT_EXCHANGE, GRANNE_HIP_SHARDED_EXCHANGE_RCCL)); }
    void set_depth(size_t depth) { check(granne_hip_sharded_set_option(h_.get(), GRANNE_HIP_SHARDED_OPT_DEPTH, depth)); }

    std::vector<std::pair<size_t, float>> search(const Element& element, size_t max_search, size_t num_neighbors) const {
        return search_batches(&element, 1, 1, max_search, num_neighbors)[0];
    }
    // n_batches batches of nq queries each (elements: n_batches * nq of them), pipelined inside the library
    std::vector<std::vector<std::pair<size_t, float>>> search_batches(const Element* elements, size_t n_batches, size_t nq,
                                                                      size_t max_search, size_t num_neighbors) const {
        const size_t total = n_batches * nq, dim = granne_hip_index_dim(granne_hip_sharded_shard(h_.get(), 0));
        std::vector<typename decltype(Element::data)::value_type> q(total * dim);
        for (size_t i = 0; i < total; ++i) {
            if (elements[i].len() != dim) throw std::runtime_error("query dimension mismatch");
            std::memcpy(q.data() + i * dim, elements[i].as_slice(), dim * sizeof(q[0]));
        }
        std::vector<uint64_t> ids(total * num_neighbors);
        std::vector<float> ds(total * num_neighbors);
        std::vector<uint32_t> counts(total);
        check(granne_hip_sharded_search_batches(h_.get(), q.data(), (uint32_t)n_batches, (uint32_t)nq, (uint32_t)max_search,
                                                (uint32_t)num_neighbors, ids.data(), ds.data(), counts.data()));
        std::vector<std::vector<std::pair<size_t, float>>> out(total);
        for (size_t i = 0; i < total; ++i)
            for (uint32_t j = 0; j < counts[i]; ++j) out[i].emplace_back((size_t)ids[i * num_neighbors + j], ds[i * num_neighbors + j]);
        return out;
    }

private:
    ShardedGranne() = default;
    std::vector<Granne<Elements>> shards_; // empty when the handle owns its shards (build)
    std::shared_ptr<granne_hip_sharded> h_;
};

// GranneBuilder (src/index/mod.rs:293-531) + the Builder trait (:303-315)
template <class Elements>
class GranneBuilder {
public:
    GranneBuilder(const BuildConfig& config, const Elements& elements, int device = 0) {
        granne_hip_builder* b = nullptr;
        using Scalar = typename decltype(Elements::Element::data)::value_type;
        check(granne_hip_builder_create(&b, &config.raw(), elements.as_slice(), elements.len(),
                                        (uint32_t)(elements.dim() ? elements.dim() : 1), detail::dtype_of<Scalar>::value, device));
        b_.reset(b, granne_hip_builder_destroy);
    }
    // GranneBuilder::from_bytes (mod.rs:430-461): a builder resuming from a written index
    static GranneBuilder from_bytes(const BuildConfig& config, const void* index, size_t index_len, const Elements& elements,
                                    int device = 0) {
        GranneBuilder b(config, elements, device);
        check(granne_hip_builder_load_index(b.b_.get(), index, index_len));
        return b;
    }
    // Builder::push (mod.rs:303-315): indexed by the next build()
    void push(const typename Elements::Element& element) {
        check(granne_hip_builder_append(b_.get(), element.as_slice(), 1));
    }
    void build() { check(granne_hip_builder_build(b_.get(), GRANNE_HIP_BUILD_ALL)); }
    void build_partial(size_t num_elements) {
        if (num_elements == 0) return; // mod.rs:375-377
        check(granne_hip_builder_build(b_.get(), num_elements));
    }
    size_t len() const { return granne_hip_builder_len(b_.get()); }
    size_t num_elements() const { return granne_hip_builder_num_elements(b_.get()); }
    size_t num_layers() const { return granne_hip_builder_num_layers(b_.get()); }
    size_t layer_len(size_t layer) const { return granne_hip_builder_layer_len(b_.get(), (uint32_t)layer); }
    Granne<Elements> get_index() const {
        granne_hip_index* h = nullptr;
        check(granne_hip_builder_get_index(b_.get(), &h));
        return Granne<Elements>(h);
    }

private:
    std::shared_ptr<granne_hip_builder> b_;
};

} // namespace granne
