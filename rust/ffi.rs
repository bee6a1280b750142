//! Rust declarations of the C ABI in `include/mrs.h` and a thin safe wrapper.
//!
//! SOURCE ONLY: the build image has no rustc/cargo, so this file has not been compiled.
//! It shows what a maintainer of the reference adds in place of its wgpu plumbing
//! (`bind_group_layouts.rs`, `shader_modules.rs`, the pipeline and bind-group code of
//! `src/main.rs:256-576`).  Link with `cargo:rustc-link-lib=dylib=mrs`.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};

pub const MRS_OK: c_int = 0;
pub const ALGO_THREE_KERNEL: i32 = 0;
pub const ALGO_ONESWEEP: i32 = 1;
pub const ALGO_HYBRID: i32 = 2;
pub const SCHEDULE_PASSES: i32 = 0;
pub const SCHEDULE_MSD: i32 = 1;
pub const SCHEDULE_MSD_REFUSED: i32 = 2;
pub const SCHEDULE_MSD_ADAPTIVE: i32 = 3;
pub const MORTON_UNIFORMS_BYTES: usize = 51248;

#[repr(C)]
pub struct mrs_context { _private: [u8; 0] }
#[repr(C)]
pub struct mrs_plan { _private: [u8; 0] }
#[repr(C)]
pub struct mrs_comm { _private: [u8; 0] }
#[repr(C)]
pub struct mrs_dist { _private: [u8; 0] }
#[repr(C)]
pub struct mrs_splitter { _private: [u8; 0] }

/// `mrs_dist_desc`: the sharded sort of one rank.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mrs_dist_desc {
    pub n_local_max: u64,
    pub key_words: i32,
    pub payload_bytes: i32,
    pub end_bit: i32,
    pub exchange_chunks: i32,
    pub tolerance: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mrs_hybrid_split {
    pub offered: i32,
    pub windowed: i32,
    pub leading_bits: u32,
    pub shift: u32,
    pub base: u32,
    pub local_bits: u32,
    pub local_capacity: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mrs_dist_info {
    pub recv_capacity: u64,
    pub device_bytes: u64,
    pub last_levels: u32,
    pub last_exact_cuts: u32,
    pub last_n_out: u64,
    pub last_ms: [f32; 4],
    pub last_sort_pass_ms: f32,
    pub last_sort_passes: u32,
    pub last_sort_schedule: [u32; 2],
    pub last_sort_piece_pairs: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mrs_split_result {
    pub world: i32,
    pub levels: i32,
    pub total: u64,
    pub rank_size: [u64; 16],
    pub cut_key: [u64; 15],
    pub exact_mask: u32,
    pub reserved: u32,
    pub send_count: [[u64; 16]; 16],
}

/// Replaces the constants of `src/constants.rs` and the sizing of `create_all_buffers`.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PlanDesc {
    pub n: u64,
    pub key_words: i32,
    pub payload_bytes: i32,
    pub digit_bits: i32,
    pub begin_bit: i32,
    pub end_bit: i32,
    pub algo: i32,
    pub tile_items: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct PlanInfo {
    pub num_passes: u32,
    pub radix: u32,
    pub tile_items: u32,
    pub threads: u32,
    pub max_tiles: u64,
    pub histogram_len: u64,
    pub temp_bytes: u64,
}

extern "C" {
    pub fn mrs_last_error_string() -> *const c_char;
    pub fn mrs_version() -> c_int;
    pub fn mrs_context_create(device_id: c_int, out: *mut *mut mrs_context) -> c_int;
    pub fn mrs_context_destroy(ctx: *mut mrs_context) -> c_int;
    pub fn mrs_context_get_attribute(ctx: *const mrs_context, attribute: c_int, value: *mut i64) -> c_int;
    pub fn mrs_context_set_attribute(ctx: *mut mrs_context, attribute: c_int, value: i64) -> c_int;
    pub fn mrs_plan_create(ctx: *mut mrs_context, desc: *const PlanDesc, out: *mut *mut mrs_plan) -> c_int;
    pub fn mrs_plan_destroy(plan: *mut mrs_plan) -> c_int;
    pub fn mrs_plan_get_info(plan: *const mrs_plan, info: *mut PlanInfo) -> c_int;
    pub fn mrs_plan_check(plan: *mut mrs_plan, stream: *mut c_void) -> c_int;
    pub fn mrs_plan_get_passes_left_out(plan: *mut mrs_plan, stream: *mut c_void, left_out_mask: *mut u32) -> c_int;
    pub fn mrs_hybrid_split_for(n: u64, begin_bit: i32, end_bit: i32, window: *const u64, out: *mut mrs_hybrid_split) -> c_int;
    pub fn mrs_plan_get_schedule(plan: *mut mrs_plan, stream: *mut c_void, schedule: *mut i32,
                                 largest_sub_bucket: *mut u32) -> c_int;
    pub fn mrs_plan_get_oversized(plan: *mut mrs_plan, stream: *mut c_void, sub_buckets: *mut u32, pairs: *mut u64,
                                  capacity: *mut u32) -> c_int;
    pub fn mrs_plan_get_adaptive(plan: *mut mrs_plan, stream: *mut c_void, buckets: *mut u32, groups: *mut u32) -> c_int;
    pub fn mrs_plan_set_timing(plan: *mut mrs_plan, enable: c_int) -> c_int;
    pub fn mrs_plan_get_timings(plan: *mut mrs_plan, cap: c_int, ms_out: *mut f32, labels_out: *mut c_char,
                                count_out: *mut c_int) -> c_int;
    pub fn mrs_morton_encode_lut(ctx: *mut mrs_context, uniforms_dev: *const c_void, vertices_dev: *const c_void,
                                 num_vertices: u64, triangles_dev: *const c_void, n: u64, codes_dev: *mut u32,
                                 stream: *mut c_void) -> c_int;
    pub fn mrs_morton_encode30(ctx: *mut mrs_context, xyz_dev: *const f32, n: u64, mn: *const f32, mx: *const f32,
                               keys_dev: *mut u32, idx_dev: *mut u32, stream: *mut c_void) -> c_int;
    pub fn mrs_sort_pairs(plan: *mut mrs_plan, n: u64, keys_a: *mut c_void, vals_a: *mut c_void,
                          keys_b: *mut c_void, vals_b: *mut c_void, stream: *mut c_void,
                          result_in_b: *mut c_int) -> c_int;
    pub fn mrs_morton_encode30_for_sort(plan: *mut mrs_plan, xyz_dev: *const f32, n: u64, mn: *const f32, mx: *const f32,
                                        keys_dev: *mut u32, idx_dev: *mut u32, stream: *mut c_void) -> c_int;
    pub fn mrs_sort_pairs_primed(plan: *mut mrs_plan, n: u64, keys_a: *mut c_void, vals_a: *mut c_void,
                                 keys_b: *mut c_void, vals_b: *mut c_void, stream: *mut c_void,
                                 result_in_b: *mut c_int) -> c_int;
    /// `mrs_sort_pairs` of (key, source index) pairs: the payload of input element i is i and is
    /// made up by the first pass; `vals_a` is ping-pong space only.
    pub fn mrs_sort_pairs_indexed(plan: *mut mrs_plan, n: u64, keys_a: *mut c_void, vals_a: *mut u32,
                                  keys_b: *mut c_void, vals_b: *mut u32, stream: *mut c_void,
                                  result_in_b: *mut c_int) -> c_int;
    pub fn mrs_sort_pairs_noalloc(ctx: *mut mrs_context, n: u64, keys_a: *mut u32, vals_a: *mut u32, keys_b: *mut u32,
                                  vals_b: *mut u32, begin_bit: i32, end_bit: i32, stream: *mut c_void,
                                  result_in_b: *mut c_int) -> c_int;
    pub fn mrs_sort_gather(plan: *mut mrs_plan, n: u64, keys_a: *mut c_void, keys_b: *mut c_void, index_a: *mut u32,
                           index_b: *mut u32, records_in: *const c_void, records_out: *mut c_void, record_bytes: u32,
                           stream: *mut c_void, result_in_b: *mut c_int) -> c_int;
    pub fn mrs_histogram(plan: *mut mrs_plan, n: u64, keys_dev: *const c_void, pass: u32, hist_dev: *mut u32,
                         stream: *mut c_void) -> c_int;
    pub fn mrs_scan(plan: *mut mrs_plan, n: u64, hist_dev: *mut u32, stream: *mut c_void) -> c_int;
    pub fn mrs_scatter(plan: *mut mrs_plan, n: u64, keys_in: *const c_void, vals_in: *const c_void,
                       keys_out: *mut c_void, vals_out: *mut c_void, scanned_hist_dev: *const u32, pass: u32,
                       stream: *mut c_void) -> c_int;
    pub fn mrs_bucket_histogram(ctx: *mut mrs_context, keys_dev: *const c_void, n: u64, key_words: i32,
                                shift: u32, bits: u32, counts_dev: *mut u64, stream: *mut c_void) -> c_int;
    pub fn mrs_partition(plan: *mut mrs_plan, n: u64, keys_in: *const c_void, vals_in: *const c_void,
                         keys_out: *mut c_void, vals_out: *mut c_void, shift: u32, bits: u32,
                         map_dev: *const u8, num_parts: u32, part_counts_dev: *mut u32,
                         stream: *mut c_void) -> c_int;
    // ---- the sort sharded over the GPUs of a node (one rank per GPU, RCCL underneath) ----
    pub fn mrs_comm_get_unique_id(id_out: *mut c_void) -> c_int; // 128 bytes, rank 0
    pub fn mrs_comm_create_rccl(unique_id: *const c_void, rank: i32, world: i32, out: *mut *mut mrs_comm) -> c_int;
    pub fn mrs_comm_wrap_rccl(nccl_comm: *mut c_void, rank: i32, world: i32, out: *mut *mut mrs_comm) -> c_int;
    pub fn mrs_comm_create_local(world: i32, out: *mut *mut mrs_comm) -> c_int; // out[0..world)
    pub fn mrs_comm_destroy(comm: *mut mrs_comm) -> c_int;
    pub fn mrs_dist_create(ctx: *mut mrs_context, comm: *mut mrs_comm, desc: *const mrs_dist_desc,
                           out: *mut *mut mrs_dist) -> c_int;
    pub fn mrs_dist_destroy(dist: *mut mrs_dist) -> c_int;
    pub fn mrs_dist_set_timing(dist: *mut mrs_dist, enable: c_int) -> c_int;
    pub fn mrs_dist_get_info(dist: *mut mrs_dist, info: *mut mrs_dist_info) -> c_int;
    pub fn mrs_dist_sort_pairs(dist: *mut mrs_dist, n_local: u64, keys_in: *const c_void, vals_in: *const c_void,
                               keys_out: *mut *mut c_void, vals_out: *mut *mut c_void, n_out: *mut u64,
                               stream: *mut c_void) -> c_int;
    pub fn mrs_dist_check(dist: *mut mrs_dist, stream: *mut c_void) -> c_int;
    pub fn mrs_splitter_create(world: i32, key_bits: i32, tolerance: u64, out: *mut *mut mrs_splitter) -> c_int;
    pub fn mrs_splitter_destroy(sp: *mut mrs_splitter) -> c_int;
    pub fn mrs_splitter_next(sp: *mut mrs_splitter, shift: *mut u32, bits: *mut u32, num_prefixes: *mut u32,
                             prefixes: *mut u64) -> c_int;
    pub fn mrs_splitter_feed(sp: *mut mrs_splitter, counts: *const u64) -> c_int;
    pub fn mrs_splitter_get(sp: *const mrs_splitter, out: *mut mrs_split_result) -> c_int;
    pub fn mrs_generate_points(ctx: *mut mrs_context, seed: u64, n: u64, xyz_dev: *mut f32,
                               stream: *mut c_void) -> c_int;
    pub fn mrs_generate_u32(ctx: *mut mrs_context, seed: u64, stream_id: u64, n: u64, out_dev: *mut u32,
                            stream: *mut c_void) -> c_int;
}

#[derive(Debug)]
pub struct Error { pub code: i32, pub message: String }
pub type Result<T> = std::result::Result<T, Error>;

fn check(rc: c_int) -> Result<()> {
    if rc == MRS_OK { return Ok(()); }
    let message = unsafe { CStr::from_ptr(mrs_last_error_string()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

pub struct Context { raw: *mut mrs_context }
impl Context {
    pub fn new(device_id: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { mrs_context_create(device_id, &mut raw) })?;
        Ok(Self { raw })
    }
    /// "Morton Coding" (src/main.rs:581-592); all pointers are device pointers.
    pub fn morton_encode_lut(&self, uniforms: *const c_void, vertices: *const c_void, num_vertices: u64,
                             triangles: *const c_void, n: u64, codes: *mut u32, stream: *mut c_void) -> Result<()> {
        check(unsafe { mrs_morton_encode_lut(self.raw, uniforms, vertices, num_vertices, triangles, n, codes, stream) })
    }
}
impl Drop for Context { fn drop(&mut self) { unsafe { mrs_context_destroy(self.raw); } } }

pub struct Plan { raw: *mut mrs_plan }
impl Plan {
    pub fn new(ctx: &Context, desc: PlanDesc) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { mrs_plan_create(ctx.raw, &desc, &mut raw) })?;
        Ok(Self { raw })
    }
    /// The NUM_PASSES loop (src/main.rs:594-665).  Returns true when the result is in the
    /// second buffer pair (odd pass count, src/main.rs:819-823).
    pub fn sort_pairs(&self, n: u64, keys_a: *mut c_void, vals_a: *mut c_void, keys_b: *mut c_void,
                      vals_b: *mut c_void, stream: *mut c_void) -> Result<bool> {
        let mut in_b: c_int = 0;
        check(unsafe { mrs_sort_pairs(self.raw, n, keys_a, vals_a, keys_b, vals_b, stream, &mut in_b) })?;
        Ok(in_b != 0)
    }
    pub fn check(&self, stream: *mut c_void) -> Result<()> { check(unsafe { mrs_plan_check(self.raw, stream) }) }
}
impl Drop for Plan { fn drop(&mut self) { unsafe { mrs_plan_destroy(self.raw); } } }
