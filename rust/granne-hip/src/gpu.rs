//! MI355X back end for `Granne::search` (feature "hip"): the index lives in HBM, the walk runs in
//! libgranne_hip.so. Results equal `Granne::search` bit for bit (ids, distances, order).
use crate::{angular, angular_int, Index};
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct granne_hip_index { _private: [u8; 0] }
#[repr(C)] pub struct granne_hip_builder { _private: [u8; 0] }
#[repr(C)] pub struct granne_hip_sharded { _private: [u8; 0] }

pub const GRANNE_HIP_F32: c_int = 0;
pub const GRANNE_HIP_I8: c_int = 1;
pub const GRANNE_HIP_BUILD_ALL: u64 = u64::MAX;

/// `granne_hip_build_config` (include/granne_hip.h) = `BuildConfig` (src/index/mod.rs:198-231)
/// + the GPU builder's batch schedule.
#[repr(C)]
pub struct granne_hip_build_config {
    pub layer_multiplier: f32,
    pub expected_num_elements: u64,
    pub num_neighbors: u32,
    pub max_search: u32,
    pub reinsert_elements: c_int,
    pub show_progress: c_int,
    pub batch_max: u32,
    pub batch_div: u32,
}

extern "C" {
    fn granne_hip_last_error() -> *const c_char;
    fn granne_hip_device_count(out_count: *mut c_int) -> c_int;
    // ---- index
    fn granne_hip_index_create(out: *mut *mut granne_hip_index, elements: *const c_void, n_elements: u64,
        dim: u32, dtype: c_int, n_layers: u32, layer_len: *const u64,
        layer_rows: *const *const u32, layer_width: *const u32, device_id: c_int) -> c_int;
    fn granne_hip_index_create_csr(out: *mut *mut granne_hip_index, elements: *const c_void, n_elements: u64,
        dim: u32, dtype: c_int, n_layers: u32, layer_len: *const u64,
        layer_offsets: *const *const u64, layer_ids: *const *const u32, device_id: c_int) -> c_int;
    // Granne::from_bytes / from_file (src/index/mod.rs:108-137): granne's own index + elements files
    fn granne_hip_index_load(out: *mut *mut granne_hip_index, index_bytes: *const c_void, index_len: u64,
        elements_bytes: *const c_void, elements_len: u64, dtype: c_int, device_id: c_int) -> c_int;
    fn granne_hip_index_load_files(out: *mut *mut granne_hip_index, index_path: *const c_char,
        elements_path: *const c_char, dtype: c_int, device_id: c_int) -> c_int;
    fn granne_hip_index_save(index: *const granne_hip_index, index_path: *const c_char, elements_path: *const c_char) -> c_int;
    fn granne_hip_index_encode(index: *const granne_hip_index, out_bytes: *mut *mut c_void, out_len: *mut u64) -> c_int;
    fn granne_hip_bytes_free(bytes: *mut c_void);
    fn granne_hip_index_destroy(index: *mut granne_hip_index);
    fn granne_hip_index_len(index: *const granne_hip_index) -> u64;
    fn granne_hip_index_num_layers(index: *const granne_hip_index) -> u32;
    fn granne_hip_index_layer_len(index: *const granne_hip_index, layer: u32) -> u64;
    fn granne_hip_index_dim(index: *const granne_hip_index) -> u32;
    fn granne_hip_index_get_neighbors(index: *const granne_hip_index, node: u64, layer: u32,
        out_ids: *mut u32, cap: u32, out_count: *mut u32) -> c_int;
    fn granne_hip_index_get_element(index: *const granne_hip_index, idx: u64, out: *mut c_void) -> c_int;
    // ---- search
    fn granne_hip_search(index: *const granne_hip_index, query: *const c_void, max_search: u32,
        num_neighbors: u32, out_ids: *mut u64, out_dists: *mut f32, out_count: *mut u32) -> c_int;
    fn granne_hip_search_batch(index: *const granne_hip_index, queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, out_ids: *mut u64, out_dists: *mut f32,
        out_counts: *mut u32, out_stats: *mut u64) -> c_int;
    // ---- Granne::reorder / reorder_by_keys (src/index/reorder.rs:59-133)
    fn granne_hip_index_reorder(index: *mut granne_hip_index, out_order: *mut u64) -> c_int;
    fn granne_hip_index_reorder_by_keys(index: *mut granne_hip_index, keys: *const u64, out_order: *mut u64) -> c_int;
    // ---- ElementContainer::dist_to_element for explicit pairs (src/elements/dense_vector.rs:149-163)
    fn granne_hip_dist_pairs(index: *const granne_hip_index, queries: *const c_void, nq: u32,
        qidx: *const u32, ids: *const u32, n_pairs: u64, out: *mut f32) -> c_int;
    // ---- GranneBuilder (src/index/mod.rs:295-531)
    fn granne_hip_build_config_default(config: *mut granne_hip_build_config);
    fn granne_hip_builder_create(out: *mut *mut granne_hip_builder, config: *const granne_hip_build_config,
        elements: *const c_void, n_elements: u64, dim: u32, dtype: c_int, device_id: c_int) -> c_int;
    fn granne_hip_builder_append(builder: *mut granne_hip_builder, elements: *const c_void, n_new: u64) -> c_int;
    fn granne_hip_builder_load_index(builder: *mut granne_hip_builder, index_bytes: *const c_void, index_len: u64) -> c_int;
    fn granne_hip_builder_build(builder: *mut granne_hip_builder, num_elements: u64) -> c_int;
    fn granne_hip_builder_len(builder: *const granne_hip_builder) -> u64;
    fn granne_hip_builder_num_layers(builder: *const granne_hip_builder) -> u32;
    fn granne_hip_builder_layer_len(builder: *const granne_hip_builder, layer: u32) -> u64;
    fn granne_hip_builder_get_layer(builder: *const granne_hip_builder, layer: u32, out_rows: *mut u32) -> c_int;
    fn granne_hip_builder_get_index(builder: *const granne_hip_builder, out: *mut *mut granne_hip_index) -> c_int;
    fn granne_hip_builder_destroy(builder: *mut granne_hip_builder);
    // ---- partitioned index, one host process (device_ids / n_devices / partitioned)
    fn granne_hip_sharded_create(out: *mut *mut granne_hip_sharded, shards: *const *mut granne_hip_index,
        id_offsets: *const u64, n_shards: u32) -> c_int;
    fn granne_hip_sharded_create_grouped(out: *mut *mut granne_hip_sharded, shards: *const *mut granne_hip_index,
        id_offsets: *const u64, n_shards: u32, groups: *const u32) -> c_int;
    fn granne_hip_sharded_destroy(sharded: *mut granne_hip_sharded);
    fn granne_hip_sharded_len(sharded: *const granne_hip_sharded) -> u64;
    fn granne_hip_sharded_search_batch(sharded: *mut granne_hip_sharded, queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, out_ids: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    fn granne_hip_sharded_search_batches(sharded: *mut granne_hip_sharded, queries: *const c_void, n_batches: u32,
        nq: u32, max_search: u32, num_neighbors: u32, out_ids: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    fn granne_hip_sharded_set_option(sharded: *mut granne_hip_sharded, option: c_int, value: u64) -> c_int;
    fn granne_hip_sharded_build(out: *mut *mut granne_hip_sharded, config: *const granne_hip_build_config,
        elements: *const c_void, n_elements: u64, dim: u32, dtype: c_int, n_shards: u32, device_ids: *const c_int,
        n_devices: u32) -> c_int;
}
pub const GRANNE_HIP_OPT_SEARCH_DEPTH: c_int = 9;
pub const GRANNE_HIP_OPT_INLINE_TAILS: c_int = 10;
pub const GRANNE_HIP_OPT_SEEN_MIN: c_int = 11;
pub const GRANNE_HIP_SHARDED_OPT_DEPTH: c_int = 1;
pub const GRANNE_HIP_SHARDED_OPT_EXCHANGE: c_int = 2;
pub const GRANNE_HIP_SHARDED_EXCHANGE_PEER: u64 = 0;
pub const GRANNE_HIP_SHARDED_EXCHANGE_RCCL: u64 = 1;

// The rest of the C ABI (include/granne_hip.h), not used by the wrappers below: generated from the header by
// tools/gen_rust_sys.py; tests/test_abi.py checks that every entry point is declared here exactly as the header has it.
extern "C" {
    // ---- library / handle queries
    fn granne_hip_abi_version() -> c_int;
    fn granne_hip_index_dtype(index: *const granne_hip_index) -> c_int;
    fn granne_hip_index_device(index: *const granne_hip_index) -> c_int;
    fn granne_hip_index_hbm_bytes(index: *const granne_hip_index) -> u64;
    fn granne_hip_index_last_slow_count(index: *const granne_hip_index) -> u64;
    fn granne_hip_index_set_option(index: *mut granne_hip_index, option: c_int, value: u64) -> c_int;
    fn granne_hip_index_get_option(index: *const granne_hip_index, option: c_int, value: *mut u64) -> c_int;
    fn granne_hip_builder_num_elements(builder: *const granne_hip_builder) -> u64;
    fn granne_hip_sharded_num_shards(sharded: *const granne_hip_sharded) -> u32;
    fn granne_hip_sharded_search(sharded: *mut granne_hip_sharded, query: *const c_void, max_search: u32,
        num_neighbors: u32, out_ids: *mut u64, out_dists: *mut f32, out_count: *mut u32) -> c_int;
    fn granne_hip_sharded_device(sharded: *const granne_hip_sharded) -> c_int;
    fn granne_hip_sharded_shard(sharded: *const granne_hip_sharded, shard: u32) -> *mut granne_hip_index;
    fn granne_hip_sharded_shard_offset(sharded: *const granne_hip_sharded, shard: u32) -> u64;
    fn granne_hip_sharded_get_option(sharded: *const granne_hip_sharded, option: c_int, value: *mut u64) -> c_int;
    // ---- Vector::from for whole arrays (src/elements/angular.rs:55-61, angular_int.rs:27-45), host buffers
    fn granne_hip_normalize_f32(rows: *mut f32, n: u64, dim: u32, device_id: c_int) -> c_int;
    fn granne_hip_quantize_f32(rows: *const f32, out: *mut i8, n: u64, dim: u32, device_id: c_int) -> c_int;
    // ---- granne's files without an index handle (src/index/io.rs:11-113, src/slice_vector/mod.rs:460-466)
    fn granne_hip_write_index_file(path: *const c_char, n_layers: u32, layer_len: *const u64,
        layer_rows: *const *const u32, layer_width: *const u32) -> c_int;
    fn granne_hip_write_elements_file(path: *const c_char, elements: *const c_void, n_elements: u64, dim: u32,
        dtype: c_int) -> c_int;
    fn granne_hip_index_file_info(index_bytes: *const c_void, index_len: u64, out_n_layers: *mut u32,
        out_layer_len: *mut u64, out_layer_ids: *mut u64, cap: u32) -> c_int;
    fn granne_hip_index_file_decode_layer(index_bytes: *const c_void, index_len: u64, layer: u32,
        out_offsets: *mut u64, out_ids: *mut u32) -> c_int;
    // ---- device-pointer entry points: for a host that already holds HIP memory and a hipStream_t (stream: *mut c_void); asynchronous
    fn granne_hip_index_create_device(out: *mut *mut granne_hip_index, d_elements: *const c_void, n_elements: u64,
        dim: u32, dtype: c_int, n_layers: u32, layer_len: *const u64, d_layer_rows: *const *const u32,
        layer_width: *const u32, device_id: c_int, stream: *mut c_void) -> c_int;
    fn granne_hip_builder_create_device(out: *mut *mut granne_hip_builder, config: *const granne_hip_build_config,
        d_elements: *const c_void, n_elements: u64, dim: u32, dtype: c_int, device_id: c_int, stream: *mut c_void) -> c_int;
    fn granne_hip_search_batch_device(index: *const granne_hip_index, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32,
        d_out_stats: *mut u64, d_status: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_search_batches_device(index: *const granne_hip_index, n_batches: u32,
        d_queries: *const *const c_void, nq: u32, max_search: u32, num_neighbors: u32, d_out_ids: *const *mut u64,
        d_out_dists: *const *mut f32, d_out_counts: *const *mut u32, d_out_stats: *const *mut u64,
        d_status: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_search_begin_device(index: *const granne_hip_index, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32,
        d_out_stats: *mut u64, d_status: *mut u32, stream: *mut c_void, out_ticket: *mut u64) -> c_int;
    fn granne_hip_search_end_device(index: *const granne_hip_index, ticket: u64, stream: *mut c_void) -> c_int;
    // ---- device memory and streams for a host without HIP bindings of its own (DeviceBuffer / Stream below)
    fn granne_hip_device_malloc(out_ptr: *mut *mut c_void, bytes: u64, device_id: c_int) -> c_int;
    fn granne_hip_device_free(ptr: *mut c_void, device_id: c_int) -> c_int;
    fn granne_hip_copy_to_device(d_dst: *mut c_void, src: *const c_void, bytes: u64, device_id: c_int,
        stream: *mut c_void) -> c_int;
    fn granne_hip_copy_to_host(dst: *mut c_void, d_src: *const c_void, bytes: u64, device_id: c_int,
        stream: *mut c_void) -> c_int;
    fn granne_hip_stream_create(out_stream: *mut *mut c_void, device_id: c_int) -> c_int;
    fn granne_hip_stream_destroy(stream: *mut c_void, device_id: c_int) -> c_int;
    fn granne_hip_stream_synchronize(stream: *mut c_void, device_id: c_int) -> c_int;
    fn granne_hip_sharded_search_batch_device(sharded: *mut granne_hip_sharded, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32,
        d_status: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_sharded_begin_device(sharded: *mut granne_hip_sharded, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32,
        d_status: *mut u32, stream: *mut c_void, out_ticket: *mut u64) -> c_int;
    fn granne_hip_sharded_end_device(sharded: *mut granne_hip_sharded, ticket: u64, stream: *mut c_void) -> c_int;
    fn granne_hip_search_batch_device_timed(index: *const granne_hip_index, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32,
        d_out_stats: *mut u64, d_status: *mut u32, stream: *mut c_void, ev_before: *mut c_void,
        ev_after: *mut c_void) -> c_int;
    fn granne_hip_search_batch_packed_device(index: *const granne_hip_index, d_queries: *const c_void, nq: u32,
        max_search: u32, num_neighbors: u32, d_packed: *mut c_void, d_status: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_packed_topk_bytes(nq: u32, k: u32) -> u64;
    fn granne_hip_brute_force_device(index: *const granne_hip_index, d_queries: *const c_void, nq: u32, k: u32,
        d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_brute_force(index: *const granne_hip_index, queries: *const c_void, nq: u32, k: u32,
        out_ids: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    fn granne_hip_merge_topk_device(d_ids: *const u64, d_dists: *const f32, d_counts: *const u32,
        shard_offsets: *const u64, n_shards: u32, nq: u32, k: u32, d_out_ids: *mut u64, d_out_dists: *mut f32,
        d_out_counts: *mut u32, device_id: c_int, stream: *mut c_void) -> c_int;
    fn granne_hip_merge_topk_packed_device(d_packed: *const c_void, shard_offsets: *const u64, n_shards: u32,
        nq: u32, k: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32, device_id: c_int,
        stream: *mut c_void) -> c_int;
    fn granne_hip_merge_topk_packed_strided_device(d_packed: *const c_void, stride_bytes: u64,
        shard_offsets: *const u64, n_shards: u32, nq: u32, k: u32, d_out_ids: *mut u64, d_out_dists: *mut f32,
        d_out_counts: *mut u32, device_id: c_int, stream: *mut c_void) -> c_int;
    fn granne_hip_dist_pairs_device(index: *const granne_hip_index, d_queries: *const c_void, d_qidx: *const u32,
        d_ids: *const u32, n_pairs: u64, d_out: *mut f32, stream: *mut c_void) -> c_int;
    fn granne_hip_dists_device(index: *const granne_hip_index, d_queries: *const c_void, nq: u32,
        d_ids: *const u32, m: u32, d_out: *mut f32, d_status: *mut u32, stream: *mut c_void) -> c_int;
    fn granne_hip_normalize_f32_device(d_rows: *mut f32, n: u64, dim: u32, device_id: c_int, stream: *mut c_void) -> c_int;
    fn granne_hip_quantize_f32_device(d_rows: *const f32, d_out: *mut i8, n: u64, dim: u32, device_id: c_int,
        stream: *mut c_void) -> c_int;
    fn granne_hip_synth_rows_device(d_out: *mut f32, seed: u64, row0: u64, n: u64, dim: u32, device_id: c_int,
        stream: *mut c_void) -> c_int;
    fn granne_hip_event_create(out_event: *mut *mut c_void) -> c_int;
    fn granne_hip_event_destroy(event: *mut c_void);
    fn granne_hip_event_elapsed_ms(before: *mut c_void, after: *mut c_void, out_ms: *mut f32) -> c_int;
}

fn check(rc: c_int) -> std::io::Result<()> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { std::ffi::CStr::from_ptr(granne_hip_last_error()) }.to_string_lossy().into_owned();
    Err(std::io::Error::new(std::io::ErrorKind::Other, format!("granne_hip {}: {}", rc, msg)))
}

pub fn device_count() -> usize {
    let mut n: c_int = 0;
    unsafe { granne_hip_device_count(&mut n) };
    n.max(0) as usize
}

/// What the two dense element containers share, as the library needs it: prepared scalars, row major.
pub trait GpuElements {
    type Vector;
    const DTYPE: c_int;
    fn scalars(&self) -> *const c_void;
    fn n(&self) -> usize;
    fn width(&self) -> usize;
    fn query_scalars(v: &Self::Vector) -> *const c_void;
    fn query_len(v: &Self::Vector) -> usize;
    fn append_query(buf: &mut Vec<u8>, v: &Self::Vector);
}
macro_rules! gpu_elements {
    ($m:ident, $scalar:ty, $dtype:expr) => {
        impl<'a> GpuElements for $m::Vectors<'a> {
            type Vector = $m::Vector<'static>;
            const DTYPE: c_int = $dtype;
            fn scalars(&self) -> *const c_void { self.as_slice().as_ptr() as *const c_void }
            fn n(&self) -> usize { self.len() }
            fn width(&self) -> usize { self.dim() }
            fn query_scalars(v: &Self::Vector) -> *const c_void { v.as_slice().as_ptr() as *const c_void }
            fn query_len(v: &Self::Vector) -> usize { v.len() }
            fn append_query(buf: &mut Vec<u8>, v: &Self::Vector) {
                let s = v.as_slice();
                buf.extend_from_slice(unsafe {
                    std::slice::from_raw_parts(s.as_ptr() as *const u8, s.len() * std::mem::size_of::<$scalar>())
                });
            }
        }
    };
}
gpu_elements!(angular, f32, GRANNE_HIP_F32);
gpu_elements!(angular_int, i8, GRANNE_HIP_I8);

/// A `Granne` whose layers and elements live in the HBM of one MI355X.
pub struct GpuGranne<E: GpuElements> { handle: *mut granne_hip_index, dim: usize, _e: PhantomData<E> }
unsafe impl<E: GpuElements> Send for GpuGranne<E> {}
unsafe impl<E: GpuElements> Sync for GpuGranne<E> {} // granne_hip_search_batch is thread-safe on a shared index

impl<E: GpuElements> GpuGranne<E> {
    fn wrap(handle: *mut granne_hip_index) -> Self {
        Self { handle, dim: unsafe { granne_hip_index_dim(handle) } as usize, _e: PhantomData }
    }
    /// From a builder's parts: `GranneBuilder::get_index()` hands out FixWidth layers (mod.rs:483-488);
    /// `layers[i]` is the row-major u32 matrix of layer i, `width` = config.num_neighbors, UNUSED padded.
    pub fn from_fix_width(elements: &E, layers: &[&[u32]], width: usize, device: i32) -> std::io::Result<Self> {
        let lens: Vec<u64> = layers.iter().map(|l| (l.len() / width) as u64).collect();
        let rows: Vec<*const u32> = layers.iter().map(|l| l.as_ptr()).collect();
        let widths: Vec<u32> = vec![width as u32; layers.len()];
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_index_create(&mut h, elements.scalars(), elements.n() as u64,
            elements.width() as u32, E::DTYPE, layers.len() as u32, lens.as_ptr(), rows.as_ptr(), widths.as_ptr(), device) })?;
        Ok(Self::wrap(h))
    }
    /// From decoded `MultiSetVector` layers (`get_into`, src/slice_vector/set_vector.rs:65-69): CSR per layer.
    pub fn from_csr(elements: &E, offsets: &[&[u64]], ids: &[&[u32]], device: i32) -> std::io::Result<Self> {
        let lens: Vec<u64> = offsets.iter().map(|o| (o.len() - 1) as u64).collect();
        let po: Vec<*const u64> = offsets.iter().map(|o| o.as_ptr()).collect();
        let pi: Vec<*const u32> = ids.iter().map(|i| i.as_ptr()).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_index_create_csr(&mut h, elements.scalars(), elements.n() as u64,
            elements.width() as u32, E::DTYPE, offsets.len() as u32, lens.as_ptr(), po.as_ptr(), pi.as_ptr(), device) })?;
        Ok(Self::wrap(h))
    }
    /// `Granne::from_bytes(index, elements)` (src/index/mod.rs:108-117) with the ELEMENTS as file bytes too
    /// (`Vectors::from_bytes`, dense_vector.rs:50-66): the library decodes granne's formats itself.
    pub fn from_bytes(index: &[u8], elements_file: &[u8], device: i32) -> std::io::Result<Self> {
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_index_load(&mut h, index.as_ptr() as *const c_void, index.len() as u64,
            elements_file.as_ptr() as *const c_void, elements_file.len() as u64, E::DTYPE, device) })?;
        Ok(Self::wrap(h))
    }
    /// `Granne::from_file` (src/index/mod.rs:120-137): the files are mapped by the library, uploaded, unmapped.
    pub fn from_files(index_path: &std::path::Path, elements_path: &std::path::Path, device: i32) -> std::io::Result<Self> {
        use std::os::unix::ffi::OsStrExt;
        let ip = std::ffi::CString::new(index_path.as_os_str().as_bytes())?;
        let ep = std::ffi::CString::new(elements_path.as_os_str().as_bytes())?;
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_index_load_files(&mut h, ip.as_ptr(), ep.as_ptr(), E::DTYPE, device) })?;
        Ok(Self::wrap(h))
    }
    /// `write_index` + `write_elements` (src/index/io.rs:11-70, slice_vector/mod.rs:460-466) to two files.
    pub fn save(&self, index_path: &std::path::Path, elements_path: &std::path::Path) -> std::io::Result<()> {
        use std::os::unix::ffi::OsStrExt;
        let ip = std::ffi::CString::new(index_path.as_os_str().as_bytes())?;
        let ep = std::ffi::CString::new(elements_path.as_os_str().as_bytes())?;
        check(unsafe { granne_hip_index_save(self.handle, ip.as_ptr(), ep.as_ptr()) })
    }

    /// Same contract as `Granne::search` (src/index/mod.rs:140-150).
    pub fn search(&self, element: &E::Vector, max_search: usize, num_neighbors: usize) -> Vec<(usize, f32)> {
        assert_eq!(E::query_len(element), self.dim);
        let (mut ids, mut ds) = (vec![0u64; num_neighbors], vec![0f32; num_neighbors]);
        let mut count = 0u32;
        // the reference panics on max_search == 0 (mod.rs:1019); the ABI returns an error code
        check(unsafe { granne_hip_search(self.handle, E::query_scalars(element), max_search as u32,
            num_neighbors as u32, ids.as_mut_ptr(), ds.as_mut_ptr(), &mut count) }).expect("granne_hip_search");
        (0..count as usize).map(|j| (ids[j] as usize, ds[j])).collect()
    }
    /// The batch form the GPU is built for: `elements.len()` independent searches in one launch.
    pub fn search_batch(&self, elements: &[E::Vector], max_search: usize, num_neighbors: usize) -> Vec<Vec<(usize, f32)>> {
        let nq = elements.len();
        let mut q = Vec::new();
        for e in elements { assert_eq!(E::query_len(e), self.dim); E::append_query(&mut q, e); }
        let (mut ids, mut ds) = (vec![0u64; nq * num_neighbors], vec![0f32; nq * num_neighbors]);
        let mut counts = vec![0u32; nq];
        check(unsafe { granne_hip_search_batch(self.handle, q.as_ptr() as *const c_void, nq as u32,
            max_search as u32, num_neighbors as u32, ids.as_mut_ptr(), ds.as_mut_ptr(),
            counts.as_mut_ptr(), std::ptr::null_mut()) }).expect("granne_hip_search_batch");
        (0..nq).map(|i| (0..counts[i] as usize)
            .map(|j| (ids[i * num_neighbors + j] as usize, ds[i * num_neighbors + j])).collect()).collect()
    }
    // ---- the device-resident entries, safe: buffers and streams are owned types (below), a batch in flight BORROWS its
    // buffers until it is ended, so the compiler keeps them alive and unaliased for exactly as long as the GPU uses them
    /// Queries of one batch in HBM (`Vector::from` applied on the host: `E::append_query`).
    pub fn upload_queries(&self, elements: &[E::Vector], stream: &Stream) -> std::io::Result<DeviceBuffer<u8>> {
        let mut q = Vec::new();
        for e in elements { assert_eq!(E::query_len(e), self.dim); E::append_query(&mut q, e); }
        let bytes = unsafe { std::slice::from_raw_parts(q.as_ptr() as *const u8, q.len() * std::mem::size_of_val(&q[0])) };
        let buf = DeviceBuffer::<u8>::new(bytes.len(), self.device())?;
        buf.copy_from(bytes, stream)?;
        stream.synchronize()?; // (`q` is dropped at the end of this function: the copy has to be over)
        Ok(buf)
    }
    pub fn device(&self) -> i32 { unsafe { granne_hip_index_device(self.handle) } }
    /// `granne_hip_search_batch_device`: one batch, one launch, ordered on `stream`.
    pub fn search_batch_device(&self, queries: &DeviceBuffer<u8>, nq: usize, max_search: usize, out: &mut DeviceResults,
                               stream: &Stream) -> std::io::Result<()> {
        out.fits(nq)?;
        check(unsafe { granne_hip_search_batch_device(self.handle, queries.ptr as *const c_void, nq as u32, max_search as u32,
            out.k as u32, out.ids.ptr, out.dists.ptr, out.counts.ptr, std::ptr::null_mut(), std::ptr::null_mut(), stream.raw) })
    }
    /// `granne_hip_search_batches_device`: several batches of `nq` queries in ONE launch (the 5.7 M f32 / 21.8 M int8
    /// queries/s path of DESIGN.md 5): walks of later batches take the places of finished ones inside the launch.
    pub fn search_batches_device(&self, queries: &[&DeviceBuffer<u8>], nq: usize, max_search: usize,
                                 out: &mut [DeviceResults], stream: &Stream) -> std::io::Result<()> {
        assert_eq!(queries.len(), out.len());
        let k = out.first().map(|o| o.k).unwrap_or(0);
        for o in out.iter() { o.fits(nq)?; assert_eq!(o.k, k); }
        let q: Vec<*const c_void> = queries.iter().map(|b| b.ptr as *const c_void).collect();
        let ids: Vec<*mut u64> = out.iter().map(|o| o.ids.ptr).collect();
        let ds: Vec<*mut f32> = out.iter().map(|o| o.dists.ptr).collect();
        let cs: Vec<*mut u32> = out.iter().map(|o| o.counts.ptr).collect();
        check(unsafe { granne_hip_search_batches_device(self.handle, q.len() as u32, q.as_ptr(), nq as u32, max_search as u32,
            k as u32, ids.as_ptr(), ds.as_ptr(), cs.as_ptr(), std::ptr::null(), std::ptr::null_mut(), stream.raw) })
    }
    /// `granne_hip_search_begin_device`: the batch runs on one of the index's own streams, ordered after what `stream`
    /// holds; the returned `InFlight` borrows the queries and the results until `end` (or its drop) has made `stream`
    /// wait for the search. Up to `set_search_depth` (default 3, at most 16) batches may be in flight.
    pub fn begin<'a>(&'a self, queries: &'a DeviceBuffer<u8>, nq: usize, max_search: usize, out: &'a mut DeviceResults,
                     stream: &'a Stream) -> std::io::Result<InFlight<'a, E>> {
        out.fits(nq)?;
        let mut ticket = 0u64;
        check(unsafe { granne_hip_search_begin_device(self.handle, queries.ptr as *const c_void, nq as u32, max_search as u32,
            out.k as u32, out.ids.ptr, out.dists.ptr, out.counts.ptr, std::ptr::null_mut(), std::ptr::null_mut(), stream.raw,
            &mut ticket) })?;
        Ok(InFlight { index: self, ticket, stream, done: false, _buffers: PhantomData })
    }
    /// `GRANNE_HIP_OPT_SEARCH_DEPTH`: how many `begin`s may be outstanding (short int8 walks want 8 and
    /// `GPU_MAX_HW_QUEUES` raised before the process's first HIP call).
    pub fn set_search_depth(&mut self, depth: u64) -> std::io::Result<()> {
        check(unsafe { granne_hip_index_set_option(self.handle, GRANNE_HIP_OPT_SEARCH_DEPTH, depth) })
    }
    /// `GRANNE_HIP_OPT_INLINE_TAILS`: 100-d / 200-d f32 indexes keep, for the walker, a copy of every layer in which a
    /// node's neighbor ids are followed by the tails of those neighbors' rows (whole-line reads; +6.9 GB of HBM at
    /// 10M x 100-d). On by default; `false` frees the copy. Results do not depend on it.
    pub fn set_inline_tails(&mut self, keep: bool) -> std::io::Result<()> {
        check(unsafe { granne_hip_index_set_option(self.handle, GRANNE_HIP_OPT_INLINE_TAILS, keep as u64) })
    }
    /// `GRANNE_HIP_OPT_SEEN_MIN`: f32 launches of at least this many walks skip a revisited neighbor BEFORE its row is
    /// fetched (a cache of the ids the walk has evaluated; a miss means nothing, so results do not depend on it). Default
    /// 2048; 0 = every launch; `u32::MAX as u64` = never.
    pub fn set_seen_min(&mut self, walks: u64) -> std::io::Result<()> {
        check(unsafe { granne_hip_index_set_option(self.handle, GRANNE_HIP_OPT_SEEN_MIN, walks) })
    }
    // ---- the Index trait's accessors (src/index/mod.rs:54-71); `impl Index for GpuGranne` below forwards to them
    pub fn len(&self) -> usize { unsafe { granne_hip_index_len(self.handle) as usize } }
    pub fn num_layers(&self) -> usize { unsafe { granne_hip_index_num_layers(self.handle) as usize } }
    pub fn layer_len(&self, layer: usize) -> usize { unsafe { granne_hip_index_layer_len(self.handle, layer as u32) as usize } }
    pub fn get_neighbors(&self, index: usize, layer: usize) -> Vec<usize> {
        let mut buf = vec![0u32; 512];
        let mut n = 0u32;
        check(unsafe { granne_hip_index_get_neighbors(self.handle, index as u64, layer as u32, buf.as_mut_ptr(),
            buf.len() as u32, &mut n) }).expect("granne_hip_index_get_neighbors");
        buf[..n as usize].iter().map(|&i| i as usize).collect()
    }
    /// `Granne::reorder` (src/index/reorder.rs:59-85): `permutation[i] == j` means the element with idx `j`
    /// has been moved to idx `i`.
    pub fn reorder(&mut self, _show_progress: bool) -> Vec<usize> {
        let mut order = vec![0u64; self.len()];
        check(unsafe { granne_hip_index_reorder(self.handle, order.as_mut_ptr()) }).expect("granne_hip_index_reorder");
        order.into_iter().map(|i| i as usize).collect()
    }
    pub fn reorder_by_keys(&mut self, keys: &[u64], _show_progress: bool) -> Vec<usize> {
        assert_eq!(self.len(), keys.len()); // reorder.rs:91
        let mut order = vec![0u64; self.len()];
        check(unsafe { granne_hip_index_reorder_by_keys(self.handle, keys.as_ptr(), order.as_mut_ptr()) })
            .expect("granne_hip_index_reorder_by_keys");
        order.into_iter().map(|i| i as usize).collect()
    }
    /// `ElementContainer::dists(&self, element, indices)` (src/elements/mod.rs:35-39)
    pub fn dists(&self, element: &E::Vector, indices: &[usize]) -> Vec<f32> {
        let ids: Vec<u32> = indices.iter().map(|&i| i as u32).collect();
        let qidx = vec![0u32; ids.len()];
        let mut out = vec![0f32; ids.len()];
        check(unsafe { granne_hip_dist_pairs(self.handle, E::query_scalars(element), 1, qidx.as_ptr(), ids.as_ptr(),
            ids.len() as u64, out.as_mut_ptr()) }).expect("granne_hip_dist_pairs");
        out
    }
}
impl<E: GpuElements> Drop for GpuGranne<E> { fn drop(&mut self) { unsafe { granne_hip_index_destroy(self.handle) } } }

/// The reference's `Index` trait (src/index/mod.rs:54-71): code that is generic over `Index` takes a `GpuGranne`.
impl<E: GpuElements> Index for GpuGranne<E> {
    fn len(self: &Self) -> usize { GpuGranne::len(self) }
    fn num_layers(self: &Self) -> usize { GpuGranne::num_layers(self) }
    fn layer_len(self: &Self, layer: usize) -> usize { GpuGranne::layer_len(self, layer) }
    fn get_neighbors(self: &Self, index: usize, layer: usize) -> Vec<usize> { GpuGranne::get_neighbors(self, index, layer) }
    /// `io::write_index` (src/index/io.rs:11-70) of the layers in HBM: the library downloads and encodes them
    /// (granne's compressed format), the bytes go to `buffer` as they are.
    fn write_index<B: std::io::Write + std::io::Seek>(self: &Self, buffer: &mut B) -> std::io::Result<()> {
        let mut bytes: *mut c_void = std::ptr::null_mut();
        let mut len = 0u64;
        check(unsafe { granne_hip_index_encode(self.handle, &mut bytes, &mut len) })?;
        let res = buffer.write_all(unsafe { std::slice::from_raw_parts(bytes as *const u8, len as usize) });
        unsafe { granne_hip_bytes_free(bytes) };
        res
    }
}

/// A batch begun with `GpuGranne::begin` and not yet ended. Ending (explicitly, or by drop) makes the caller's stream
/// wait for the search; the borrow of the query and result buffers ends with this value.
pub struct InFlight<'a, E: GpuElements> {
    index: &'a GpuGranne<E>, ticket: u64, stream: &'a Stream, done: bool,
    _buffers: PhantomData<(&'a DeviceBuffer<u8>, &'a mut DeviceResults)>,
}
impl<'a, E: GpuElements> InFlight<'a, E> {
    pub fn end(mut self) -> std::io::Result<()> { self.finish() }
    fn finish(&mut self) -> std::io::Result<()> {
        if self.done { return Ok(()); }
        self.done = true;
        check(unsafe { granne_hip_search_end_device(self.index.handle, self.ticket, self.stream.raw) })
    }
}
impl<'a, E: GpuElements> Drop for InFlight<'a, E> { fn drop(&mut self) { let _ = self.finish(); } }

/// `hipStream_t`, created and destroyed through the library (a host with HIP bindings of its own wraps its stream with
/// `Stream::borrowed` instead).
pub struct Stream { raw: *mut c_void, device: i32, owned: bool }
unsafe impl Send for Stream {}
impl Stream {
    pub fn new(device: i32) -> std::io::Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { granne_hip_stream_create(&mut raw, device) })?;
        Ok(Self { raw, device, owned: true })
    }
    /// # Safety: `raw` is a live hipStream_t of `device` for as long as the value exists.
    pub unsafe fn borrowed(raw: *mut c_void, device: i32) -> Self { Self { raw, device, owned: false } }
    pub fn synchronize(&self) -> std::io::Result<()> { check(unsafe { granne_hip_stream_synchronize(self.raw, self.device) }) }
}
impl Drop for Stream {
    fn drop(&mut self) { if self.owned { unsafe { granne_hip_stream_synchronize(self.raw, self.device); granne_hip_stream_destroy(self.raw, self.device); } } }
}

/// `len` values of `T` in the HBM of one device.
pub struct DeviceBuffer<T: Copy> { ptr: *mut T, len: usize, device: i32 }
unsafe impl<T: Copy> Send for DeviceBuffer<T> {}
impl<T: Copy> DeviceBuffer<T> {
    pub fn new(len: usize, device: i32) -> std::io::Result<Self> {
        let mut p = std::ptr::null_mut();
        check(unsafe { granne_hip_device_malloc(&mut p, (len * std::mem::size_of::<T>()) as u64, device) })?;
        Ok(Self { ptr: p as *mut T, len, device })
    }
    pub fn len(&self) -> usize { self.len }
    /// host -> device, ordered on `stream`; `src` must stay alive until the stream has passed the copy (synchronize, or
    /// keep it in a structure that outlives the work: pageable memory is staged by the runtime before the call returns)
    pub fn copy_from(&self, src: &[T], stream: &Stream) -> std::io::Result<()> {
        assert!(src.len() <= self.len);
        check(unsafe { granne_hip_copy_to_device(self.ptr as *mut c_void, src.as_ptr() as *const c_void,
            (src.len() * std::mem::size_of::<T>()) as u64, self.device, stream.raw) })
    }
    /// device -> host; returns after `stream` has been waited for (the data is in `dst`)
    pub fn copy_to(&self, dst: &mut [T], stream: &Stream) -> std::io::Result<()> {
        assert!(dst.len() <= self.len);
        check(unsafe { granne_hip_copy_to_host(dst.as_mut_ptr() as *mut c_void, self.ptr as *const c_void,
            (dst.len() * std::mem::size_of::<T>()) as u64, self.device, stream.raw) })?;
        stream.synchronize()
    }
}
impl<T: Copy> Drop for DeviceBuffer<T> { fn drop(&mut self) { unsafe { granne_hip_device_free(self.ptr as *mut c_void, self.device); } } }

/// The results of one batch in HBM: `[nq][k]` ids and distances, `[nq]` counts.
pub struct DeviceResults { pub ids: DeviceBuffer<u64>, pub dists: DeviceBuffer<f32>, pub counts: DeviceBuffer<u32>, nq: usize, k: usize }
impl DeviceResults {
    pub fn new(nq: usize, k: usize, device: i32) -> std::io::Result<Self> {
        Ok(Self { ids: DeviceBuffer::new(nq * k, device)?, dists: DeviceBuffer::new(nq * k, device)?,
                  counts: DeviceBuffer::new(nq, device)?, nq, k })
    }
    fn fits(&self, nq: usize) -> std::io::Result<()> {
        if nq <= self.nq { Ok(()) } else { Err(std::io::Error::new(std::io::ErrorKind::InvalidInput, "result buffers are smaller than the batch")) }
    }
    /// `Vec<Vec<(usize, f32)>>` as `Granne::search` returns it, one per query (waits for `stream`)
    pub fn to_host(&self, nq: usize, stream: &Stream) -> std::io::Result<Vec<Vec<(usize, f32)>>> {
        let (mut ids, mut ds, mut cs) = (vec![0u64; nq * self.k], vec![0f32; nq * self.k], vec![0u32; nq]);
        self.ids.copy_to(&mut ids, stream)?;
        self.dists.copy_to(&mut ds, stream)?;
        self.counts.copy_to(&mut cs, stream)?;
        Ok((0..nq).map(|i| (0..cs[i] as usize).map(|j| (ids[i * self.k + j] as usize, ds[i * self.k + j])).collect()).collect())
    }
}

pub type GpuAngularGranne<'a> = GpuGranne<angular::Vectors<'a>>;
pub type GpuAngularIntGranne<'a> = GpuGranne<angular_int::Vectors<'a>>;

/// `GranneBuilder` on the GPU: the same layer pyramid and per-element work, a deterministic batched
/// insertion schedule instead of rayon + per-node locks.
pub struct GpuGranneBuilder<E: GpuElements> { handle: *mut granne_hip_builder, _e: PhantomData<E> }
unsafe impl<E: GpuElements> Send for GpuGranneBuilder<E> {}

/// `BuildConfig` (src/index/mod.rs:198-283) for the GPU builder: the same builder-style methods, the same defaults.
/// `BuildConfig`'s own fields are private to `index` and it has no getters (mod.rs:199-214), so a back end outside
/// that module cannot read one; inside the granne crate a maintainer adds `pub(crate)` getters and
/// `impl From<BuildConfig> for GpuBuildConfig`, and every `GpuBuildConfig` below can be spelled `BuildConfig`.
#[derive(Clone, Debug)]
pub struct GpuBuildConfig {
    layer_multiplier: f32,
    expected_num_elements: Option<usize>,
    num_neighbors: usize,
    max_search: usize,
    reinsert_elements: bool,
    show_progress: bool,
}
impl Default for GpuBuildConfig {
    /// mod.rs:217-231
    fn default() -> Self {
        Self { layer_multiplier: 15.0, expected_num_elements: None, num_neighbors: 30, max_search: 200,
               reinsert_elements: true, show_progress: false }
    }
}
impl GpuBuildConfig {
    pub fn new() -> Self { Self::default() }
    pub fn layer_multiplier(mut self, layer_multiplier: f32) -> Self { self.layer_multiplier = layer_multiplier; self }
    pub fn expected_num_elements(mut self, n: usize) -> Self { self.expected_num_elements = Some(n); self }
    pub fn num_neighbors(mut self, num_neighbors: usize) -> Self { self.num_neighbors = num_neighbors; self }
    pub fn max_search(mut self, max_search: usize) -> Self { self.max_search = max_search; self }
    pub fn reinsert_elements(mut self, yes: bool) -> Self { self.reinsert_elements = yes; self }
    pub fn show_progress(mut self, yes: bool) -> Self { self.show_progress = yes; self }
}

fn to_c(config: &GpuBuildConfig) -> granne_hip_build_config {
    // the batch schedule (batch_max, batch_div) keeps the library's defaults
    let mut c = std::mem::MaybeUninit::<granne_hip_build_config>::uninit();
    unsafe { granne_hip_build_config_default(c.as_mut_ptr()) };
    let mut c = unsafe { c.assume_init() };
    c.layer_multiplier = config.layer_multiplier;
    c.expected_num_elements = config.expected_num_elements.unwrap_or(0) as u64;
    c.num_neighbors = config.num_neighbors as u32;
    c.max_search = config.max_search as u32;
    c.reinsert_elements = config.reinsert_elements as c_int;
    c.show_progress = config.show_progress as c_int;
    c
}

impl<E: GpuElements> GpuGranneBuilder<E> {
    /// `GranneBuilder::new(config, elements)` (mod.rs:419-428)
    pub fn new(config: GpuBuildConfig, elements: &E, device: i32) -> std::io::Result<Self> {
        let c = to_c(&config);
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_builder_create(&mut h, &c, elements.scalars(), elements.n() as u64,
            elements.width() as u32, E::DTYPE, device) })?;
        Ok(Self { handle: h, _e: PhantomData })
    }
    /// `GranneBuilder::from_bytes(config, buffer, elements)` (mod.rs:430-461): continue from a written index
    pub fn from_bytes(config: GpuBuildConfig, buffer: &[u8], elements: &E, device: i32) -> std::io::Result<Self> {
        let b = Self::new(config, elements, device)?;
        check(unsafe { granne_hip_builder_load_index(b.handle, buffer.as_ptr() as *const c_void, buffer.len() as u64) })?;
        Ok(b)
    }
    /// `Builder::push` for prepared rows (mod.rs:303-315, 527-531)
    pub fn extend(&mut self, elements: &E) -> std::io::Result<()> {
        check(unsafe { granne_hip_builder_append(self.handle, elements.scalars(), elements.n() as u64) })
    }
    pub fn build(&mut self) { check(unsafe { granne_hip_builder_build(self.handle, GRANNE_HIP_BUILD_ALL) }).expect("build") }
    pub fn build_partial(&mut self, num_elements: usize) {
        check(unsafe { granne_hip_builder_build(self.handle, num_elements as u64) }).expect("build_partial") // 0: no-op, mod.rs:375
    }
    pub fn len(&self) -> usize { unsafe { granne_hip_builder_len(self.handle) as usize } }
    pub fn num_layers(&self) -> usize { unsafe { granne_hip_builder_num_layers(self.handle) as usize } }
    pub fn layer_len(&self, layer: usize) -> usize { unsafe { granne_hip_builder_layer_len(self.handle, layer as u32) as usize } }
    /// layer `layer` as a row-major [layer_len][num_neighbors] u32 matrix, UNUSED padded (host copy)
    pub fn layer_rows(&self, layer: usize, num_neighbors: usize) -> Vec<u32> {
        let mut rows = vec![0u32; self.layer_len(layer) * num_neighbors];
        check(unsafe { granne_hip_builder_get_layer(self.handle, layer as u32, rows.as_mut_ptr()) }).expect("get_layer");
        rows
    }
    /// `GranneBuilder::get_index()` (mod.rs:483-488): a searchable index over the builder's current layers
    pub fn get_index(&self) -> GpuGranne<E> {
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_builder_get_index(self.handle, &mut h) }).expect("get_index");
        GpuGranne::wrap(h)
    }
}
impl<E: GpuElements> Drop for GpuGranneBuilder<E> { fn drop(&mut self) { unsafe { granne_hip_builder_destroy(self.handle) } } }

/// A partitioned index: shard `s` is a `GpuGranne` of its own (on the device it was created on) over the
/// elements `[offsets[s], offsets[s] + shard.len())` of the whole set -- how the reference's own shard helper
/// cuts an element file (src/elements/embeddings/parsing.rs:63-100). `search` asks every shard and merges by
/// (distance, global id); ids in the result are global.
pub struct GpuShardedGranne<E: GpuElements> { handle: *mut granne_hip_sharded, shards: Vec<GpuGranne<E>>, dim: usize }
unsafe impl<E: GpuElements> Send for GpuShardedGranne<E> {}

impl<E: GpuElements> GpuShardedGranne<E> {
    pub fn new(shards: Vec<GpuGranne<E>>, offsets: &[u64]) -> std::io::Result<Self> {
        assert_eq!(shards.len(), offsets.len());
        let hs: Vec<*mut granne_hip_index> = shards.iter().map(|s| s.handle).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_sharded_create(&mut h, hs.as_ptr(), offsets.as_ptr(), hs.len() as u32) })?;
        let dim = shards[0].dim;
        Ok(Self { handle: h, shards, dim })
    }
    /// The whole element set in, a searchable partitioned index out: `n_shards` id ranges
    /// (src/elements/embeddings/parsing.rs:72-98), each built with the GPU builder under `config` on
    /// `devices[s / ceil(n_shards / devices.len())]`. The handle owns its shards (`shards` stays empty).
    pub fn build(config: GpuBuildConfig, elements: &E, n_shards: usize, devices: &[i32]) -> std::io::Result<Self> {
        let cfg = to_c(&config);
        let mut h = std::ptr::null_mut();
        check(unsafe { granne_hip_sharded_build(&mut h, &cfg, elements.scalars(), elements.n() as u64,
            elements.width() as u32, E::DTYPE, n_shards as u32, devices.as_ptr(), devices.len() as u32) })?;
        Ok(Self { handle: h, shards: Vec::new(), dim: elements.width() })
    }
    pub fn len(&self) -> usize { unsafe { granne_hip_sharded_len(self.handle) as usize } }
    pub fn num_shards(&self) -> usize { unsafe { granne_hip_sharded_num_shards(self.handle) as usize } }
    pub fn search(&mut self, element: &E::Vector, max_search: usize, num_neighbors: usize) -> Vec<(usize, f32)> {
        self.search_batch(std::slice::from_ref(element), max_search, num_neighbors).pop().unwrap()
    }
    pub fn search_batch(&mut self, elements: &[E::Vector], max_search: usize, num_neighbors: usize) -> Vec<Vec<(usize, f32)>> {
        let nq = elements.len();
        let mut q = Vec::new();
        for e in elements { assert_eq!(E::query_len(e), self.dim); E::append_query(&mut q, e); }
        let (mut ids, mut ds) = (vec![0u64; nq * num_neighbors], vec![0f32; nq * num_neighbors]);
        let mut counts = vec![0u32; nq];
        check(unsafe { granne_hip_sharded_search_batch(self.handle, q.as_ptr() as *const c_void, nq as u32,
            max_search as u32, num_neighbors as u32, ids.as_mut_ptr(), ds.as_mut_ptr(), counts.as_mut_ptr()) })
            .expect("granne_hip_sharded_search_batch");
        (0..nq).map(|i| (0..counts[i] as usize)
            .map(|j| (ids[i * num_neighbors + j] as usize, ds[i * num_neighbors + j])).collect()).collect()
    }
    /// Many batches of `nq` queries each: the library pipelines them (batch b+1 is uploaded and searched while
    /// batch b is exchanged, merged and downloaded; pinned staging inside). The caller-side loop over
    /// `search_batch`, without the bubbles.
    pub fn search_batches(&mut self, elements: &[E::Vector], nq: usize, max_search: usize, num_neighbors: usize)
        -> Vec<Vec<(usize, f32)>> {
        assert!(nq > 0 && elements.len() % nq == 0);
        let (total, nb) = (elements.len(), elements.len() / nq);
        let mut q = Vec::new();
        for e in elements { assert_eq!(E::query_len(e), self.dim); E::append_query(&mut q, e); }
        let (mut ids, mut ds) = (vec![0u64; total * num_neighbors], vec![0f32; total * num_neighbors]);
        let mut counts = vec![0u32; total];
        check(unsafe { granne_hip_sharded_search_batches(self.handle, q.as_ptr() as *const c_void, nb as u32, nq as u32,
            max_search as u32, num_neighbors as u32, ids.as_mut_ptr(), ds.as_mut_ptr(), counts.as_mut_ptr()) })
            .expect("granne_hip_sharded_search_batches");
        (0..total).map(|i| (0..counts[i] as usize)
            .map(|j| (ids[i * num_neighbors + j] as usize, ds[i * num_neighbors + j])).collect()).collect()
    }
    /// `granne_hip_sharded_search_batch_device`: queries and results in the HBM of `self.device()` (shard 0's), ordered on
    /// `stream`; every shard searches the batch, the merged top-k lands in `out`.
    pub fn search_batch_device(&mut self, queries: &DeviceBuffer<u8>, nq: usize, max_search: usize, out: &mut DeviceResults,
                               stream: &Stream) -> std::io::Result<()> {
        out.fits(nq)?;
        check(unsafe { granne_hip_sharded_search_batch_device(self.handle, queries.ptr as *const c_void, nq as u32,
            max_search as u32, out.k as u32, out.ids.ptr, out.dists.ptr, out.counts.ptr, std::ptr::null_mut(), stream.raw) })
    }
    pub fn device(&self) -> i32 { unsafe { granne_hip_sharded_device(self.handle) } }
    /// `granne_hip_sharded_begin_device` / `_end_device`: batch b + 1 is searched while batch b is exchanged and merged.
    /// The ticket's borrow keeps the buffers alive; `ShardedInFlight::end` (or drop) orders `stream` after the merge.
    pub fn begin<'a>(&'a self, queries: &'a DeviceBuffer<u8>, nq: usize, max_search: usize, out: &'a mut DeviceResults,
                     stream: &'a Stream) -> std::io::Result<ShardedInFlight<'a, E>> {
        out.fits(nq)?;
        let mut ticket = 0u64;
        check(unsafe { granne_hip_sharded_begin_device(self.handle, queries.ptr as *const c_void, nq as u32, max_search as u32,
            out.k as u32, out.ids.ptr, out.dists.ptr, out.counts.ptr, std::ptr::null_mut(), stream.raw, &mut ticket) })?;
        Ok(ShardedInFlight { index: self, ticket, stream, done: false, _buffers: PhantomData })
    }
    /// The exchange step as ONE RCCL all-gather over the shard devices (instead of peer copies); batches in flight.
    pub fn use_rccl_all_gather(&mut self) -> std::io::Result<()> {
        check(unsafe { granne_hip_sharded_set_option(self.handle, GRANNE_HIP_SHARDED_OPT_EXCHANGE, GRANNE_HIP_SHARDED_EXCHANGE_RCCL) })
    }
    pub fn set_depth(&mut self, depth: u64) -> std::io::Result<()> {
        check(unsafe { granne_hip_sharded_set_option(self.handle, GRANNE_HIP_SHARDED_OPT_DEPTH, depth) })
    }
}
impl<E: GpuElements> Drop for GpuShardedGranne<E> {
    fn drop(&mut self) { unsafe { granne_hip_sharded_destroy(self.handle) } } // before `shards`: the handle borrows them
}
pub struct ShardedInFlight<'a, E: GpuElements> {
    index: &'a GpuShardedGranne<E>, ticket: u64, stream: &'a Stream, done: bool,
    _buffers: PhantomData<(&'a DeviceBuffer<u8>, &'a mut DeviceResults)>,
}
impl<'a, E: GpuElements> ShardedInFlight<'a, E> {
    pub fn end(mut self) -> std::io::Result<()> { self.finish() }
    fn finish(&mut self) -> std::io::Result<()> {
        if self.done { return Ok(()); }
        self.done = true;
        check(unsafe { granne_hip_sharded_end_device(self.index.handle, self.ticket, self.stream.raw) })
    }
}
impl<'a, E: GpuElements> Drop for ShardedInFlight<'a, E> { fn drop(&mut self) { let _ = self.finish(); } }
