//! `granne-hip`: granne's `Granne::search` path on one (or eight) AMD MI355X.
//!
//! ```ignore
//! use granne_hip::gpu::{GpuAngularGranne, GpuGranneBuilder};
//! let mut builder = GpuGranneBuilder::new(granne::BuildConfig::default(), &elements, 0)?;
//! builder.build();
//! let index = builder.get_index();              // stays in HBM
//! let res = index.search(&query, 200, 10);      // == granne::Granne::search on the same graph, bit for bit
//! ```
//!
//! `gpu.rs` is written as a module of the granne crate (`crate::angular`, `crate::BuildConfig`, ...):
//! the names it reaches through `crate::` are re-exported here so that the same file serves both places.
pub use granne::{angular, angular_int, BuildConfig, Index};

pub mod gpu;
