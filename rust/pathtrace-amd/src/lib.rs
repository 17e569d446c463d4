//! Safe surface over `pathtrace-amd-sys`.  What the reference's render thread (`src/main.rs:42-60`) needs:
//! describe the `World` once, render, get the two film buffers back (`World.data`, `World.luminance_data`,
//! `src/world.rs:55-57`).  All `unsafe` of the integration is in this file.
use pathtrace_amd_sys as sys;
use std::ffi::CStr;

pub use sys::{PtCamera as Camera, PtObject as Object, PtRenderParams as RenderParams, PtStats as Stats, PtTuning as Tuning};

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "pathtrace_amd error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(code: i32) -> Result<(), Error> {
    if code == sys::PT_OK {
        return Ok(());
    }
    // pt_last_error() is thread-local and valid until the next failing call on this thread
    let message = unsafe { CStr::from_ptr(sys::pt_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code, message })
}

/// `Object::new(Box<dyn Shape>, Box<dyn Material>)` flattened (`src/objects/object.rs:17-24`).
pub fn sphere(center: [f64; 3], radius: f64, mat_tag: u32, mat: [f64; 6]) -> Object {
    Object { shape_tag: sys::PT_SHAPE_SPHERE, mat_tag, shape: [center[0], center[1], center[2], radius, 0., 0., 0., 0., 0.], mat }
}
pub fn triangle(v0: [f64; 3], v1: [f64; 3], v2: [f64; 3], mat_tag: u32, mat: [f64; 6]) -> Object {
    Object { shape_tag: sys::PT_SHAPE_TRIANGLE, mat_tag, shape: [v0[0], v0[1], v0[2], v1[0], v1[1], v1[2], v2[0], v2[1], v2[2]], mat }
}

/// Reference constants (`world.rs:18`, `rendering.rs:6-7`): spp 3000, depth 4..50, MIS, t_min 1e-3.
pub fn default_params() -> RenderParams {
    let mut p = std::mem::MaybeUninit::<RenderParams>::zeroed();
    unsafe {
        sys::pt_default_params(p.as_mut_ptr());
        p.assume_init()
    }
}

/// `Camera::new` (`src/camera.rs:50-82`).
pub fn camera_new(origin: [f64; 3], width: u32, height: u32, screen_distance: f64, fov_degrees: f64) -> Result<Camera, Error> {
    let mut cam = Camera::default();
    check(unsafe { sys::pt_camera_new(origin.as_ptr(), width, height, screen_distance, fov_degrees, &mut cam) })?;
    Ok(cam)
}

/// The two buffers `World::render_pixel` fills (`world.rs:318-332`), for one tile.
pub struct Film {
    pub width: u32,
    pub rows: u32,
    /// mean linear radiance, RGB f32, row-major (= `luminance_data`)
    pub linear_rgb: Vec<f32>,
    /// sqrt-gamma, truncated RGBA8 (= `World.data`, what `draw()` blits)
    pub rgba8: Vec<u8>,
}

/// One GPU context with one uploaded scene.  Not `Sync`: use from one thread at a time (the reference's
/// render thread); create one per GPU for multi-GPU band rendering.
pub struct Renderer {
    ctx: *mut sys::PtContext,
}
unsafe impl Send for Renderer {}

/// The library must be the ABI version these declarations were written for (struct sizes differ between versions:
/// `PtStats` grew in version 4).  Checked before any context exists.
fn check_abi() -> Result<(), Error> {
    let lib = unsafe { sys::pt_abi_version() };
    if lib != sys::PT_ABI_VERSION {
        return Err(Error { code: sys::PT_ERR_UNSUPPORTED as i32, message: format!("pathtrace-amd-sys declares ABI version {}, the library is version {}", sys::PT_ABI_VERSION, lib) });
    }
    Ok(())
}

impl Renderer {
    pub fn new(device: i32) -> Result<Self, Error> {
        check_abi()?;
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::pt_context_create(device, &mut ctx) })?;
        Ok(Renderer { ctx })
    }
    /// Copies the scene (the reference's `World` is immutable while rendering, `world.rs:293`).
    pub fn upload(&mut self, objects: &[Object]) -> Result<(), Error> {
        check(unsafe { sys::pt_scene_upload(self.ctx, objects.as_ptr(), objects.len() as u32) })
    }
    /// Everything `main.rs:43-60` does for the tile selected by `params.band_*`; blocking.
    pub fn render(&mut self, cam: &Camera, params: &RenderParams) -> Result<Film, Error> {
        let band_count = params.band_count.max(1);
        let rows = unsafe { sys::pt_tile_rows(cam.height, params.band_rows, params.band_index, band_count) };
        let n = rows as usize * cam.width as usize;
        let mut film = Film { width: cam.width, rows, linear_rgb: vec![0f32; n * 3], rgba8: vec![0u8; n * 4] };
        check(unsafe { sys::pt_render_host(self.ctx, cam, params, film.linear_rgb.as_mut_ptr(), film.rgba8.as_mut_ptr()) })?;
        Ok(film)
    }
    /// Progressive preview (`main.rs:79-90`): `on_frame(spp_done, spp_total, rgba8)` after every `spp_step`
    /// samples; return `true` to stop.  The final film equals `render()`'s.
    pub fn render_progressive<F: FnMut(u32, u32, &[u8]) -> bool>(
        &mut self, cam: &Camera, params: &RenderParams, spp_step: u32, mut on_frame: F,
    ) -> Result<Film, Error> {
        struct Ctx<'a> {
            f: &'a mut dyn FnMut(u32, u32, &[u8]) -> bool,
            len: usize,
            panic: Option<Box<dyn std::any::Any + Send + 'static>>,
        }
        // A panic must not unwind through the C frames of pt_render_progressive (undefined behaviour): it is caught
        // here, the render is asked to stop (non-zero return), and the panic resumes on the Rust side of the call.
        unsafe extern "C" fn tramp(user: *mut std::os::raw::c_void, done: u32, total: u32, rgba8: *const u8, _lin: *const f32) -> i32 {
            let c = &mut *(user as *mut Ctx);
            let px = std::slice::from_raw_parts(rgba8, c.len);
            let f = &mut c.f;
            match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| f(done, total, px))) {
                Ok(stop) => stop as i32,
                Err(payload) => { c.panic = Some(payload); 1 }
            }
        }
        let band_count = params.band_count.max(1);
        let rows = unsafe { sys::pt_tile_rows(cam.height, params.band_rows, params.band_index, band_count) };
        let n = rows as usize * cam.width as usize;
        let mut film = Film { width: cam.width, rows, linear_rgb: vec![0f32; n * 3], rgba8: vec![0u8; n * 4] };
        let mut c = Ctx { f: &mut on_frame, len: n * 4, panic: None };
        let rc = unsafe {
            sys::pt_render_progressive(self.ctx, cam, params, spp_step, Some(tramp), &mut c as *mut Ctx as *mut _, film.linear_rgb.as_mut_ptr(), film.rgba8.as_mut_ptr())
        };
        if let Some(payload) = c.panic.take() { std::panic::resume_unwind(payload); }
        check(rc)?;
        Ok(film)
    }
    pub fn stats(&mut self) -> Result<Stats, Error> {
        let mut s = Stats::default();
        check(unsafe { sys::pt_get_stats(self.ctx, &mut s) })?;
        Ok(s)
    }
    /// Scheduling knobs (`pt_context_set_tuning`; `Tuning::default()` = the library's choices).  The film and the
    /// counters never depend on them: they exist for measurements (which kernel form a launch takes, grid sizes).
    pub fn set_tuning(&mut self, tuning: &Tuning) -> Result<(), Error> {
        check(unsafe { sys::pt_context_set_tuning(self.ctx, tuning) })
    }
    /// `World::render_pixel` (`world.rs:293-333`) for a pixel list: every listed pixel gets exactly the samples a
    /// full render gives it.  `want_samples`: also the radiance of every camera sample, `[pixel][sample][rgb]`
    /// (= `ray_color`'s return values, what the reference's diagnostics print, `world.rs:378-417`).
    pub fn render_pixels(&mut self, cam: &Camera, params: &RenderParams, pixels: &[(u32, u32)], want_samples: bool) -> Result<PixelFilm, Error> {
        let n = pixels.len();
        let xy: Vec<u32> = pixels.iter().flat_map(|&(x, y)| [x, y]).collect();
        let mut out = PixelFilm {
            linear_rgb: vec![0f32; n * 3],
            rgba8: vec![0u8; n * 4],
            samples: if want_samples { vec![0f32; n * params.spp as usize * 3] } else { Vec::new() },
        };
        let smp = if want_samples { out.samples.as_mut_ptr() } else { std::ptr::null_mut() };
        check(unsafe { sys::pt_render_pixels(self.ctx, cam, params, xy.as_ptr(), n as u32, out.linear_rgb.as_mut_ptr(), out.rgba8.as_mut_ptr(), smp) })?;
        Ok(out)
    }
    /// `RenderingStrategy::ray_color(world, ray, 0, rng, Vector3::one())` (`rendering.rs:34-142`, `214-265`) for
    /// arbitrary rays (origin, direction), the rng of ray i being the stream of key `keys[i]` at sample
    /// `params.spp_offset`.  Returns RGB per ray.
    pub fn ray_color(&mut self, params: &RenderParams, rays: &[([f64; 3], [f64; 3])], keys: &[(u32, u32)]) -> Result<Vec<[f32; 3]>, Error> {
        assert_eq!(rays.len(), keys.len());
        let r6: Vec<f64> = rays.iter().flat_map(|(o, d)| [o[0], o[1], o[2], d[0], d[1], d[2]]).collect();
        let xy: Vec<u32> = keys.iter().flat_map(|&(x, y)| [x, y]).collect();
        let mut out = vec![0f32; rays.len() * 3];
        check(unsafe { sys::pt_ray_color(self.ctx, params, r6.as_ptr(), xy.as_ptr(), rays.len() as u32, out.as_mut_ptr()) })?;
        Ok(out.chunks_exact(3).map(|c| [c[0], c[1], c[2]]).collect())
    }
}

/// Result of `Renderer::render_pixels`.
pub struct PixelFilm {
    pub linear_rgb: Vec<f32>,
    pub rgba8: Vec<u8>,
    /// `[pixel][sample][rgb]`, empty unless requested
    pub samples: Vec<f32>,
}

/// Several GPUs of one node in ONE process (`pt_multi_*`): interleaved row bands, one RCCL gather of the film to
/// the first device.  The frame does not depend on the number of devices.
pub struct Multi {
    m: *mut sys::PtMulti,
}
unsafe impl Send for Multi {}

impl Multi {
    pub fn new(devices: &[i32]) -> Result<Self, Error> {
        check_abi()?;
        let mut m = std::ptr::null_mut();
        check(unsafe { sys::pt_multi_create(devices.as_ptr(), devices.len() as u32, &mut m) })?;
        Ok(Multi { m })
    }
    pub fn upload(&mut self, objects: &[Object]) -> Result<(), Error> {
        check(unsafe { sys::pt_multi_scene_upload(self.m, objects.as_ptr(), objects.len() as u32) })
    }
    /// The whole frame (band_* of `params` are ignored: the object owns the partition); blocking.
    pub fn render(&mut self, cam: &Camera, params: &RenderParams) -> Result<Film, Error> {
        let n = cam.width as usize * cam.height as usize;
        let mut film = Film { width: cam.width, rows: cam.height, linear_rgb: vec![0f32; n * 3], rgba8: vec![0u8; n * 4] };
        check(unsafe { sys::pt_multi_render_host(self.m, cam, params, film.linear_rgb.as_mut_ptr(), film.rgba8.as_mut_ptr()) })?;
        Ok(film)
    }
    /// One host thread per device inside the library (`true`) or the calling thread for all of them, the gather calls in one
    /// `ncclGroup` (`false`, the default): same frame either way.
    pub fn set_threads(&mut self, enabled: bool) -> Result<(), Error> {
        check(unsafe { sys::pt_multi_set_threads(self.m, enabled as i32) })
    }

    /// How the tiles reach the first device: one `ncclGather` per frame (`false`, the default) or one copy per device by
    /// the DMA engines (`true`: no kernel takes part in the exchange).  Same frame either way.
    pub fn set_exchange_by_copies(&mut self, copies: bool) -> Result<(), Error> {
        check(unsafe { sys::pt_multi_set_exchange(self.m, if copies { sys::PT_EXCHANGE_COPY } else { sys::PT_EXCHANGE_RCCL }) })
    }
    /// Devices, `ncclCommCount`, RCCL version, frames posted and what a frame costs the host.
    pub fn info(&mut self) -> Result<sys::PtMultiInfo, Error> {
        let mut i = sys::PtMultiInfo::default();
        check(unsafe { sys::pt_multi_info(self.m, &mut i) })?;
        Ok(i)
    }
    /// Counters of the frames since the last collection, summed over the devices.
    pub fn stats(&mut self) -> Result<Stats, Error> {
        let mut s = Stats::default();
        check(unsafe { sys::pt_multi_get_stats(self.m, &mut s) })?;
        Ok(s)
    }
}

impl Drop for Multi {
    fn drop(&mut self) {
        unsafe { sys::pt_multi_destroy(self.m) };
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { sys::pt_context_destroy(self.ctx) };
    }
}
