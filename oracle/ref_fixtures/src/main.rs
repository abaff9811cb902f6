//! Golden-fixture emitter: runs the REFERENCE (granne 0.5.2) on seeded synthetic data and dumps what the
//! parity tests compare against -- prepared elements, the index file, and `Granne::search` results as
//! (id, distance bits). The data generator is the benchmark's counter-based one (splitmix64; identical
//! to `gro_synth_rows` in oracle/granne_oracle.c and to `synth_rows_kernel` on the device), so every
//! side regenerates the same rows from a seed.
//!
//!     granne-ref-fixtures <out_dir>
//!
//! Layout of <out_dir>/<case>/ (all little endian):
//!   manifest.json       the case parameters and the list of result files
//!   elements.bin        `write_elements` (src/slice_vector/mod.rs:460-466): [u64 dim][rows]
//!   index.granne        `write_index` (src/index/io.rs:11-70)
//!   queries.bin         [nq][dim] PREPARED query scalars (f32 normalised / i8 quantised)
//!   search_ms<M>_k<K>.bin   per query: [u32 count][count x (u64 id, u32 distance bits)]
//!   dists.bin           [nq] u32: distance bits of query i to element i (`Dist::dist`)
//!   reorder_order.bin   (cases with `reorder`) [n] u64: the permutation `Granne::reorder` returned
//!   reordered_elements.bin, reordered_search_ms<M>_k<K>.bin   the elements and the same searches after it
//! Cases with `distinct < n` repeat rows (row i is synthetic row i % distinct): exactly equal distances, the
//! tie-breaking comparisons of search_for_neighbors / MaxSizeHeap::push decide.
use granne::{angular, angular_int, BuildConfig, Builder, GranneBuilder, Index};
use std::fs::{create_dir_all, File};
use std::io::{BufWriter, Write};
use std::path::Path;

const SEED: u64 = 0x6772616e6e65; // "granne": elements; queries use SEED + 1

fn splitmix64(z0: u64) -> u64 {
    let mut z = z0.wrapping_add(0x9E3779B97F4A7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}

/// component `col` of row `row`: uniform in [-0.5, 0.5) with 24 random bits (src/test_helper.rs:3-6's range)
fn synth_row(seed: u64, row: u64, dim: usize) -> Vec<f32> {
    let s = splitmix64(seed);
    (0..dim)
        .map(|c| {
            let z = splitmix64(s ^ (row * dim as u64 + c as u64));
            ((z >> 40) as u32) as f32 * (1.0f32 / 16777216.0f32) - 0.5f32
        })
        .collect()
}

struct Case {
    name: &'static str,
    n: usize,
    dim: usize,
    nq: usize,
    num_neighbors: usize,
    build_max_search: usize,
    searches: &'static [(usize, usize)], // (max_search, num_neighbors)
    distinct: usize,                     // row i is synthetic row i % distinct (== n: all rows differ)
    reorder: bool,                       // also run Granne::reorder and search again
}

const CASES: &[Case] = &[
    Case { name: "f32_d100", n: 3000, dim: 100, nq: 64, num_neighbors: 30, build_max_search: 50, searches: &[(1, 1), (50, 10), (200, 50), (300, 300), (1024, 10)], distinct: 3000, reorder: true },
    Case { name: "f32_d28", n: 700, dim: 28, nq: 32, num_neighbors: 20, build_max_search: 30, searches: &[(5, 5), (40, 10)], distinct: 700, reorder: false },
    Case { name: "f32_d200", n: 2000, dim: 200, nq: 32, num_neighbors: 30, build_max_search: 40, searches: &[(50, 10)], distinct: 2000, reorder: false },
    // every vector three times: ties everywhere (the walk's strict / non-strict comparisons, (dist, id) order)
    Case { name: "ties_d32", n: 1500, dim: 32, nq: 48, num_neighbors: 20, build_max_search: 30, searches: &[(1, 1), (20, 10), (60, 60), (130, 20)], distinct: 500, reorder: false },
];

fn write_results<W: Write>(w: &mut W, res: &[(usize, f32)]) -> std::io::Result<()> {
    w.write_all(&(res.len() as u32).to_le_bytes())?;
    for &(id, d) in res {
        w.write_all(&(id as u64).to_le_bytes())?;
        w.write_all(&d.to_bits().to_le_bytes())?;
    }
    Ok(())
}

macro_rules! emit_case {
    ($module:ident, $scalar:ty, $tag:expr, $case:expr, $out:expr) => {{
        let case: &Case = $case;
        let dir = Path::new($out).join(format!("{}{}", case.name, $tag));
        create_dir_all(&dir)?;
        // elements: Vector::from(Vec<f32>) normalises (angular.rs:55-61) / quantises (angular_int.rs:19-45)
        let mut elements = $module::Vectors::new();
        for i in 0..case.n {
            let v: $module::Vector = synth_row(SEED, (i % case.distinct) as u64, case.dim).into();
            elements.push(&v);
        }
        let queries: Vec<$module::Vector> = (0..case.nq).map(|i| synth_row(SEED + 1, i as u64, case.dim).into()).collect();

        let config = BuildConfig::default()
            .num_neighbors(case.num_neighbors)
            .max_search(case.build_max_search)
            .show_progress(false);
        let mut builder = GranneBuilder::new(config, elements);
        builder.build();

        builder.write_elements(&mut BufWriter::new(File::create(dir.join("elements.bin"))?))?;
        builder.write_index(&mut File::create(dir.join("index.granne"))?)?;
        {
            let mut w = BufWriter::new(File::create(dir.join("queries.bin"))?);
            for q in &queries {
                for x in q.as_slice() {
                    w.write_all(&x.to_le_bytes())?;
                }
            }
        }
        let index = builder.get_index();
        let mut files = Vec::new();
        for &(ms, k) in case.searches {
            let name = format!("search_ms{}_k{}.bin", ms, k);
            let mut w = BufWriter::new(File::create(dir.join(&name))?);
            for q in &queries {
                write_results(&mut w, &index.search(q, ms, k))?; // Granne::search, src/index/mod.rs:140-150
            }
            files.push(format!("{{\"file\":\"{}\",\"max_search\":{},\"num_neighbors\":{}}}", name, ms, k));
        }
        {
            use granne::Dist;
            let mut w = BufWriter::new(File::create(dir.join("dists.bin"))?);
            for (i, q) in queries.iter().enumerate() {
                let d: f32 = index.get_element(i).dist(q).into_inner();
                w.write_all(&d.to_bits().to_le_bytes())?;
            }
        }
        let mut reorder_files = Vec::new();
        if case.reorder {
            // Granne::reorder (src/index/reorder.rs:59-85) on an owned copy: the permutation, the permuted elements and
            // the same searches afterwards (equal to the ones before modulo the permutation, reorder.rs:297-323)
            let mut owned = index.to_owned();
            let order = owned.reorder(false);
            let mut w = BufWriter::new(File::create(dir.join("reorder_order.bin"))?);
            for &o in &order {
                w.write_all(&(o as u64).to_le_bytes())?;
            }
            owned.write_elements(&mut BufWriter::new(File::create(dir.join("reordered_elements.bin"))?))?;
            for &(ms, k) in case.searches {
                let name = format!("reordered_search_ms{}_k{}.bin", ms, k);
                let mut w = BufWriter::new(File::create(dir.join(&name))?);
                for q in &queries {
                    write_results(&mut w, &owned.search(q, ms, k))?;
                }
                reorder_files.push(format!("{{\"file\":\"{}\",\"max_search\":{},\"num_neighbors\":{}}}", name, ms, k));
            }
        }
        let layers: Vec<String> = (0..index.num_layers()).map(|l| index.layer_len(l).to_string()).collect();
        let manifest = format!(
            "{{\"case\":\"{}{}\",\"element_type\":\"{}\",\"n\":{},\"dim\":{},\"nq\":{},\"seed\":{},\"num_neighbors\":{},\
             \"build_max_search\":{},\"reinsert_elements\":true,\"layer_multiplier\":15.0,\"feature\":\"singlethreaded\",\
             \"granne_version\":\"0.5.2\",\"layer_lens\":[{}],\"searches\":[{}],\"distinct\":{},\"reordered_searches\":[{}]}}\n",
            case.name, $tag, stringify!($module), case.n, case.dim, case.nq, SEED, case.num_neighbors, case.build_max_search,
            layers.join(","), files.join(","), case.distinct, reorder_files.join(",")
        );
        File::create(dir.join("manifest.json"))?.write_all(manifest.as_bytes())?;
        eprintln!("wrote {}", dir.display());
    }};
}

fn main() -> std::io::Result<()> {
    let out = std::env::args().nth(1).expect("usage: granne-ref-fixtures <out_dir>");
    for case in CASES {
        emit_case!(angular, f32, "", case, &out);
        emit_case!(angular_int, i8, "_i8", case, &out);
    }
    Ok(())
}
