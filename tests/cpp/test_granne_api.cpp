// The reference's own index tests, written against the C++ mirror of its API (include/granne.hpp).
// Runs on an MI355X (tests/test_gpu_cpp_api.py compiles and executes it).
//   build_and_search_float / _int8      /root/reference/src/index/tests.rs:41-48, 114-132
//   with_borrowed_elements (config)     src/index/tests.rs:64-82
//   incremental build_partial           src/index/tests.rs:134-168
//   write_and_load                      src/index/tests.rs:337-394
#include <fstream>
#include <iterator>
#include <algorithm>
#include <cstdio>
#include <random>

#include "granne.hpp"

using namespace granne;

static std::mt19937 rng(12345);
static std::vector<float> random_floats(size_t dim) { // src/test_helper.rs:3-6
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<float> v(dim);
    for (auto& x : v) x = u(rng) - 0.5f;
    return v;
}

template <class Elements, class From>
static Elements random_vectors(size_t dim, size_t num, From from) { // src/test_helper.rs:12-19
    Elements e;
    for (size_t i = 0; i < num; ++i) e.push(from(random_floats(dim)));
    return e;
}

template <class Elements>
static double verify_search(const Granne<Elements>& index, size_t max_search) { // src/index/tests.rs:50-62
    size_t found = 0;
    for (size_t i = 0; i < index.len(); ++i)
        if (index.search(index.get_element(i), max_search, 1)[0].first == i) ++found;
    return (double)found / (double)index.len();
}

#define REQUIRE(c)                                                         \
    do {                                                                   \
        if (!(c)) {                                                        \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    {   // build_and_search_float
        auto elements = random_vectors<angular::Vectors>(28, 1500, [](std::vector<float> v) { return angular::from(std::move(v)); });
        GranneBuilder<angular::Vectors> builder(BuildConfig().num_neighbors(20).max_search(20), elements);
        builder.build();
        REQUIRE(builder.len() == 1500 && builder.num_elements() == 1500);
        auto index = builder.get_index();
        double p1 = verify_search(index, 10);
        std::printf("build_and_search_float p1 = %.3f\n", p1);
        REQUIRE(p1 > 0.95);
        // results are sorted by (dist, id) and at most num_neighbors long
        auto r = index.search(index.get_element(7), 30, 5);
        REQUIRE(r.size() == 5);
        for (size_t i = 1; i < r.size(); ++i) REQUIRE(r[i - 1].second <= r[i].second);
        // the reference panics on max_search == 0 (src/index/mod.rs:1019)
        bool threw = false;
        try { index.search(index.get_element(0), 0, 1); } catch (const std::runtime_error&) { threw = true; }
        REQUIRE(threw);
        // the exact scan (ElementContainer::dists over every index + a sort): a member finds itself first, and the walk's
        // best hit at a generous max_search is never better than the scan's
        {
            auto e7 = index.get_element(7);
            auto exact = index.brute_force(&e7, 1, 5);
            REQUIRE(exact.size() == 1 && exact[0].size() == 5);
            REQUIRE(exact[0][0].first == 7);
            for (size_t i = 1; i < exact[0].size(); ++i) REQUIRE(exact[0][i - 1].second <= exact[0][i].second);
            auto walk = index.search(e7, 200, 5);
            REQUIRE(exact[0][0].second <= walk[0].second);
        }
        // write_and_load
        index.write_index(tmp + "/cpp_index.granne");
        index.write_elements(tmp + "/cpp_elements.bin");
        auto loaded = Granne<angular::Vectors>::from_file(tmp + "/cpp_index.granne", tmp + "/cpp_elements.bin");
        REQUIRE(loaded.len() == index.len() && loaded.num_layers() == index.num_layers());
        for (size_t i = 0; i < 50; ++i) {
            auto a = index.search(index.get_element(i * 3), 20, 5), b = loaded.search(index.get_element(i * 3), 20, 5);
            REQUIRE(a == b);
        }
        // Index::write_index into a writer (src/index/mod.rs:67-70): the same bytes as the file; the walkers' options change no result
        {
            auto blob = index.index_bytes();
            std::ifstream f(tmp + "/cpp_index.granne", std::ios::binary);
            std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            REQUIRE(blob == file && blob.size() > 1024);
            auto before = index.search(index.get_element(9), 20, 5);
            index.set_inline_tails(false);
            index.set_seen_min(0);
            REQUIRE(index.search(index.get_element(9), 20, 5) == before);
            index.set_inline_tails(true);
            index.set_seen_min(2048);
        }
    }
    {   // reorder_index (src/index/reorder.rs:299-322)
        auto elements = random_vectors<angular::Vectors>(5, 5000, [](std::vector<float> v) { return angular::from(std::move(v)); });
        GranneBuilder<angular::Vectors> builder(BuildConfig().max_search(5).layer_multiplier(5.0f), elements);
        builder.build();
        auto index = builder.get_index();
        auto reordered_index = builder.get_index();
        auto permutation = reordered_index.reorder(false);
        REQUIRE(permutation.size() == 5000);
        for (size_t idx : {0, 10, 123, 99, 499}) {
            auto element = index.get_element(idx);
            auto exp = index.search(element, 10, 10);
            auto res = reordered_index.search(element, 10, 10);
            REQUIRE(exp.size() == 10 && res.size() == 10);
            for (size_t i = 0; i < 10; ++i) REQUIRE(exp[i].first == permutation[res[i].first]);
        }
    }
    {   // build_and_search_int8
        auto elements = random_vectors<angular_int::Vectors>(32, 500, [](std::vector<float> v) { return angular_int::from(v); });
        GranneBuilder<angular_int::Vectors> builder(BuildConfig().num_neighbors(20).max_search(20), elements);
        builder.build();
        double p1 = verify_search(builder.get_index(), 10);
        std::printf("build_and_search_int8 p1 = %.3f\n", p1);
        REQUIRE(p1 > 0.95);
    }
    {   // with_borrowed_elements configuration + incremental build
        auto elements = random_vectors<angular::Vectors>(25, 500, [](std::vector<float> v) { return angular::from(std::move(v)); });
        GranneBuilder<angular::Vectors> builder(BuildConfig().max_search(5).reinsert_elements(false), elements);
        builder.build_partial(0);
        REQUIRE(builder.len() == 0);
        builder.build_partial(100);
        REQUIRE(builder.len() == 100);
        builder.build();
        REQUIRE(builder.len() == 500);
        double p1 = verify_search(builder.get_index(), 40);
        std::printf("with_borrowed_elements p1 = %.3f\n", p1);
        REQUIRE(p1 > 0.95);
    }
    {   // a partitioned index (src/elements/embeddings/parsing.rs:63-100): three shards of consecutive elements, every
        // shard asked, best by (distance, global id) -- against the per-shard searches merged here; then the same with
        // the RCCL all-gather as the exchange step (this process has no PyTorch: the system's librccl comes in by dlopen)
        const size_t n = 1800, dim = 24, G = 3, per = n / G;
        auto all = random_vectors<angular::Vectors>(dim, n, [](std::vector<float> v) { return angular::from(std::move(v)); });
        std::vector<Granne<angular::Vectors>> shards;
        std::vector<uint64_t> offsets;
        for (size_t g = 0; g < G; ++g) {
            angular::Vectors part;
            for (size_t i = g * per; i < (g + 1) * per; ++i) part.push(all.get_element(i));
            GranneBuilder<angular::Vectors> b(BuildConfig().num_neighbors(16).max_search(30), part);
            b.build();
            shards.push_back(b.get_index());
            offsets.push_back(g * per);
        }
        std::vector<angular::Vector> queries;
        for (size_t i = 0; i < 40; ++i) queries.push_back(angular::from(random_floats(dim)));
        std::vector<std::vector<std::pair<size_t, float>>> want(queries.size());
        for (size_t i = 0; i < queries.size(); ++i) {
            std::vector<std::pair<float, size_t>> cand;
            for (size_t g = 0; g < G; ++g)
                for (auto& r : shards[g].search(queries[i], 40, 5)) cand.emplace_back(r.second, r.first + offsets[g]);
            std::sort(cand.begin(), cand.end());
            for (size_t j = 0; j < 5 && j < cand.size(); ++j) want[i].emplace_back(cand[j].second, cand[j].first);
        }
        ShardedGranne<angular::Vectors> sharded(shards, offsets);
        REQUIRE(sharded.len() == n && sharded.num_shards() == G);
        for (int pass = 0; pass < 2; ++pass) {
            auto got = sharded.search_batches(queries.data(), 4, 10, 40, 5); // four batches of ten, pipelined
            REQUIRE(got.size() == queries.size());
            for (size_t i = 0; i < queries.size(); ++i) {
                REQUIRE(got[i].size() == want[i].size());
                for (size_t j = 0; j < got[i].size(); ++j)
                    REQUIRE(got[i][j].first == want[i][j].first && got[i][j].second == want[i][j].second);
            }
            auto one = sharded.search(queries[3], 40, 5);
            REQUIRE(one.size() == want[3].size() && one[0].first == want[3][0].first);
            if (pass == 0) sharded.use_rccl_all_gather();
        }
        // granne_hip_sharded_build: the same element set, split and built in one call (the same builder, the same ranges)
        auto built = ShardedGranne<angular::Vectors>::build(BuildConfig().num_neighbors(16).max_search(30), all, G);
        REQUIRE(built.len() == n && built.num_shards() == G);
        auto again = built.search_batches(queries.data(), 4, 10, 40, 5);
        for (size_t i = 0; i < queries.size(); ++i) {
            REQUIRE(again[i].size() == want[i].size());
            for (size_t j = 0; j < again[i].size(); ++j)
                REQUIRE(again[i][j].first == want[i][j].first && again[i][j].second == want[i][j].second);
        }
        std::printf("sharded: 3 shards, peer copies and RCCL all-gather agree with the merged per-shard searches; sharded_build too\n");
    }
    std::printf("ok\n");
    return 0;
}
