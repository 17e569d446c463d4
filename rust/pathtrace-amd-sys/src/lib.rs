//! Raw declarations of `include/pathtrace_amd.h` (ABI version 4).  Field order, types and names follow the
//! header exactly; `tests/test_rust_binding.py` checks that.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const PT_ABI_VERSION: u32 = 4;

pub const PT_OK: c_int = 0;
pub const PT_ERR_INVALID_ARG: c_int = 1;
pub const PT_ERR_NO_DEVICE: c_int = 2;
pub const PT_ERR_HIP: c_int = 3;
pub const PT_ERR_OOM: c_int = 4;
pub const PT_ERR_UNSUPPORTED: c_int = 5;

pub const PT_SHAPE_SPHERE: u32 = 0;
pub const PT_SHAPE_TRIANGLE: u32 = 1;
pub const PT_MAT_LAMBERT: u32 = 0;
pub const PT_MAT_EMISSIVE: u32 = 1;
pub const PT_MAT_MIRROR: u32 = 2;
pub const PT_MAT_OREN_NAYAR: u32 = 3;
pub const PT_INTEGRATOR_MIS: u32 = 0;
pub const PT_INTEGRATOR_BRDF_ONLY: u32 = 1;
pub const PT_ACCEL_LINEAR: u32 = 0;
pub const PT_ACCEL_BVH: u32 = 1;
pub const PT_ACCEL_AUTO: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtCamera {
    pub origin: [f64; 3],
    pub lower_left: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtObject {
    pub shape_tag: u32,
    pub mat_tag: u32,
    pub shape: [f64; 9],
    pub mat: [f64; 6],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtRenderParams {
    pub spp: u32,
    pub spp_offset: u32,
    pub min_depth: u32,
    pub max_depth: u32,
    pub integrator: u32,
    pub t_min: f64,
    pub band_rows: u32,
    pub band_index: u32,
    pub band_count: u32,
    pub max_paths_in_flight: u64,
    pub profile: u32,
    pub workgroups: u32,
    pub exact_math: u32,
    pub accel: u32,
    pub n_devices: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtTuning {
    pub export_below: u32,
    pub bvh_refill: u32,
    pub bvh_leaf: u32,
    pub cont_workgroups: u32,
    pub level0_form: u32,
    pub regen_workgroups: u32,
    pub in_order: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtStats {
    pub samples: u64,
    pub vertices: u64,
    pub shadow_rays: u64,
    pub bounce_launches: u32,
    pub batches: u32,
    pub max_depth_reached: u32,
    pub reserved: u32,
    pub bounce_kernel_ms: f64,
    pub total_ms: f64,
    pub primary_vertices: u64,
    pub primary_kernel_ms: f64,
    pub primary_launches: u32,
    pub reserved2: u32,
    pub samples_expected: u64,
}

/// `pt_multi_info`: what a multi-device object is made of.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtMultiInfo {
    pub n_devices: u32,
    pub comm_count: u32,
    pub rccl_version: u32,
    pub threaded: u32,
    pub frames: u64,
    pub enqueue_us_sum: f64,
    pub enqueue_us_max: f64,
    pub exchange: u32,
    pub reserved_: u32,
}
pub const PT_EXCHANGE_RCCL: u32 = 0;
pub const PT_EXCHANGE_COPY: u32 = 1;

#[repr(C)]
pub struct PtContext {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PtMulti {
    _private: [u8; 0],
}
/// The launch scheduler on a state of its own (host only; `pt_debug_sched_*`, tests).
#[repr(C)]
pub struct PtSched {
    _private: [u8; 0],
}
/// A render as the scheduler sees it (`csrc/pt_sched.h`: `Job`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtSchedJob {
    pub n_batches: u32,
    pub regen: u32,
    pub split: u32,
    pub hand_off: u32,
    pub regen_export: u32,
    pub profile: u32,
    pub in_order: u32,
    pub capturing: u32,
    pub grid: u32,
    pub regen_grid: u32,
    pub cont_grid: u32,
    pub regen_capacity: u32,
    pub fixed_grid: u32,
    pub counter_words: u32,
    pub xchg_need: u64,
}
/// One stream operation of a planned render (`csrc/pt_sched.h`: `Op`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtSchedOp {
    pub kind: u32,
    pub stream: u32,
    pub event: u32,
    pub pool: u32,
    pub set: u32,
    pub lane: u32,
    pub level: u32,
    pub own_queue: u32,
    pub ovf_par: u32,
    pub batch: u32,
    pub grid: u32,
    pub seq: u32,
    pub core: u32,
    pub flags: u32,
    pub zero_words: u32,
    pub reserved: u32,
    pub xchg_off: u64,
    pub xchg_len: u64,
}

/// `pt_context_set_stream`: HIP's legacy default stream (its handle, 0, means "the context's own stream").
pub const PT_STREAM_LEGACY_DEFAULT: usize = 1;

pub type PtProgressFn = Option<
    unsafe extern "C" fn(user: *mut c_void, spp_done: u32, spp_total: u32, rgba8: *const u8, linear_rgb: *const f32) -> c_int,
>;

extern "C" {
    pub fn pt_camera_new(origin: *const f64, width: u32, height: u32, screen_distance: f64, fov_degrees: f64, out: *mut PtCamera) -> c_int;
    pub fn pt_camera_look_at(origin: *const f64, target: *const f64, up: *const f64, width: u32, height: u32, fov_degrees: f64, out: *mut PtCamera) -> c_int;
    pub fn pt_default_params(out: *mut PtRenderParams);
    pub fn pt_tile_rows(height: u32, band_rows: u32, band_index: u32, band_count: u32) -> u32;
    pub fn pt_builtin_scene(id: u32, arg: u32, objs: *mut PtObject, cap: u32, n: *mut u32) -> c_int;
    pub fn pt_context_create(device: c_int, out: *mut *mut PtContext) -> c_int;
    pub fn pt_context_destroy(ctx: *mut PtContext) -> c_int;
    pub fn pt_context_set_stream(ctx: *mut PtContext, hip_stream: *mut c_void) -> c_int;
    pub fn pt_context_set_tuning(ctx: *mut PtContext, tuning: *const PtTuning) -> c_int;
    pub fn pt_scene_upload(ctx: *mut PtContext, objs: *const PtObject, n_objs: u32) -> c_int;
    pub fn pt_render_device(ctx: *mut PtContext, cam: *const PtCamera, params: *const PtRenderParams, d_linear_rgb: *mut f32, d_rgba8: *mut u8) -> c_int;
    pub fn pt_render_device_packed(ctx: *mut PtContext, cam: *const PtCamera, params: *const PtRenderParams, d_packed: *mut c_void) -> c_int;
    pub fn pt_sync(ctx: *mut PtContext) -> c_int;
    pub fn pt_get_stats(ctx: *mut PtContext, out: *mut PtStats) -> c_int;
    pub fn pt_debug_raw_stats(ctx: *mut PtContext, out16: *mut u64) -> c_int;
    pub fn pt_debug_scan_layout(ctx: *mut PtContext, n_spheres: *mut u32, n_triangles: *mut u32, n_pairs: *mut u32) -> c_int;
    pub fn pt_render_host(ctx: *mut PtContext, cam: *const PtCamera, params: *const PtRenderParams, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_render_progressive(ctx: *mut PtContext, cam: *const PtCamera, params: *const PtRenderParams, spp_step: u32, f: PtProgressFn, user: *mut c_void, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_render(cam: *const PtCamera, objs: *const PtObject, n_objs: u32, params: *const PtRenderParams, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_shutdown();
    pub fn pt_multi_create(devices: *const c_int, n_devices: u32, out: *mut *mut PtMulti) -> c_int;
    pub fn pt_multi_destroy(m: *mut PtMulti) -> c_int;
    pub fn pt_multi_device_count(m: *const PtMulti) -> u32;
    pub fn pt_multi_set_threads(m: *mut PtMulti, enabled: c_int) -> c_int;
    pub fn pt_multi_set_exchange(m: *mut PtMulti, mode: u32) -> c_int;
    pub fn pt_multi_info(m: *mut PtMulti, out: *mut PtMultiInfo) -> c_int;
    pub fn pt_debug_multi_create_shared(device: c_int, n: u32, out: *mut *mut PtMulti) -> c_int;
    pub fn pt_debug_feeder_selftest(n_workers: u32, n_frames: u32, spin: u32, fail_at: i32, order_out: *mut u64, n_out: *mut u32) -> c_int;
    pub fn pt_debug_sched_create(out: *mut *mut PtSched) -> c_int;
    pub fn pt_debug_sched_destroy(s: *mut PtSched);
    pub fn pt_debug_sched_render(s: *mut PtSched, job: *const PtSchedJob, faults: u32, fail_after: u32, ops: *mut PtSchedOp, cap: u32, n_ops: *mut u32, lanes: *mut u32) -> c_int;
    pub fn pt_debug_sched_sync(s: *mut PtSched, collect: u32) -> c_int;
    pub fn pt_debug_fail_after(ctx: *mut PtContext, n: i64) -> c_int;
    pub fn pt_multi_scene_upload(m: *mut PtMulti, objs: *const PtObject, n_objs: u32) -> c_int;
    pub fn pt_multi_set_tuning(m: *mut PtMulti, tuning: *const PtTuning) -> c_int;
    pub fn pt_multi_render_device(m: *mut PtMulti, cam: *const PtCamera, params: *const PtRenderParams, d_linear_rgb: *mut f32, d_rgba8: *mut u8) -> c_int;
    pub fn pt_multi_sync(m: *mut PtMulti) -> c_int;
    pub fn pt_multi_get_stats(m: *mut PtMulti, out: *mut PtStats) -> c_int;
    pub fn pt_multi_render_host(m: *mut PtMulti, cam: *const PtCamera, params: *const PtRenderParams, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_render_multi(devices: *const c_int, n_devices: u32, cam: *const PtCamera, objs: *const PtObject, n_objs: u32, params: *const PtRenderParams, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_film_pack(hip_stream: *mut c_void, d_linear_rgb: *const f32, d_rgba8: *const u8, n_pixels: u32, d_packed: *mut c_void) -> c_int;
    pub fn pt_film_unpack(hip_stream: *mut c_void, d_gathered: *const c_void, width: u32, height: u32, band_rows: u32, n_ranks: u32, max_rows: u32, d_linear_rgb: *mut f32, d_rgba8: *mut u8) -> c_int;
    pub fn pt_debug_multi_emulate(ctx: *mut PtContext, n_virtual: u32, cam: *const PtCamera, params: *const PtRenderParams, out_linear_rgb: *mut f32, out_rgba8: *mut u8) -> c_int;
    pub fn pt_render_pixels(ctx: *mut PtContext, cam: *const PtCamera, params: *const PtRenderParams, xy: *const u32, n: u32, out_linear_rgb: *mut f32, out_rgba8: *mut u8, out_samples: *mut f32) -> c_int;
    pub fn pt_ray_color(ctx: *mut PtContext, params: *const PtRenderParams, rays: *const f64, xy: *const u32, n: u32, out_rgb: *mut f32) -> c_int;
    pub fn pt_debug_hit_scene(ctx: *mut PtContext, rays: *const f64, n: u32, t_min: f64, t_max: f64, exact_math: u32, accel: u32, out_id: *mut i32, out_t: *mut f32) -> c_int;
    pub fn pt_debug_hit_records(ctx: *mut PtContext, rays: *const f64, n: u32, t_min: f64, t_max: f64, exact_math: u32, accel: u32, out_id: *mut i32, out_rec: *mut f32) -> c_int;
    pub fn pt_debug_bsdf_eval(ctx: *mut PtContext, obj: u32, in10: *const f64, n: u32, exact_math: u32, out4: *mut f32) -> c_int;
    pub fn pt_debug_bsdf_sample(ctx: *mut PtContext, obj: u32, in7: *const f64, words4: *const u32, n: u32, exact_math: u32, out8: *mut f32) -> c_int;
    pub fn pt_debug_shape_sample(ctx: *mut PtContext, obj: u32, from3: *const f64, target3: *const f64, r12: *const f64, n: u32, exact_math: u32, out8: *mut f32) -> c_int;
    pub fn pt_debug_light_point(ctx: *mut PtContext, from3: *const f64, words4: *const u32, n: u32, exact_math: u32, out8: *mut f32) -> c_int;
    pub fn pt_debug_camera_rays(ctx: *mut PtContext, cam: *const PtCamera, xys: *const u32, n: u32, exact_math: u32, out8: *mut f32) -> c_int;
    pub fn pt_debug_bvh_check(objs: *const PtObject, n_objs: u32, depth: *mut u32, n_nodes: *mut u32, n_leaf_slots: *mut u32) -> c_int;
    pub fn pt_last_error() -> *const c_char;
    pub fn pt_abi_version() -> u32;
}
