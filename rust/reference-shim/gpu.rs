//! `src/gpu.rs` -- the GPU render path of roxas1533/pathtrace (added by `reference-gpu.patch`).
//!
//! Replaces the rayon loop of `main()` (`src/main.rs:43-60`): `World` is flattened once into the POD the
//! C ABI takes (`include/pathtrace_amd.h`), rendered by `libpathtrace_amd.so` on the MI355X, and the two film
//! buffers the rest of the program reads -- `World.data` (`src/world.rs:55`, blitted by `draw()`) and
//! `World.luminance_data` (`src/world.rs:57`, written out by `export_luminance()`) -- are filled exactly as
//! `render_pixel` fills them (`src/world.rs:318-332`).  The crate keeps `#![forbid(unsafe_code)]`
//! (`src/main.rs:1`): every `unsafe` block of the integration is inside the `pathtrace-amd` wrapper crate.
//!
//! Same per-pixel seeding convention as `main.rs:51` (the RNG key of pixel (x, y) is the pair (x, y)); the
//! generator is the library's counter-based Philox4x32 (7 rounds), not rand's ChaCha12, so a GPU film and a CPU film
//! are two draws of the same estimator, not the same numbers.
use crate::math::Vector3;
use crate::world::{Color, World, HEIGHT, SAMPLE_NUM, WIDTH};
use pathtrace_amd as pt;

/// Shape / material tags of `PtObject` (`include/pathtrace_amd.h`).
pub const SHAPE_SPHERE: u32 = 0;
pub const SHAPE_TRIANGLE: u32 = 1;
pub const MAT_LAMBERT: u32 = 0;
pub const MAT_EMISSIVE: u32 = 1;
pub const MAT_MIRROR: u32 = 2;
pub const MAT_OREN_NAYAR: u32 = 3;

/// `World.objects` in order (the order decides closest-hit ties, `src/world.rs:281-287`), each
/// `Box<dyn Shape>` / `Box<dyn Material>` described by the `describe()` method the patch adds to the traits.
pub fn flatten(world: &World) -> Vec<pt::Object> {
    world
        .objects()
        .iter()
        .map(|o| {
            let (shape_tag, shape) = o.shape.describe();
            let (mat_tag, mat) = o.material.describe();
            pt::Object { shape_tag, mat_tag, shape, mat }
        })
        .collect()
}

/// The reference's compile-time constants as `PtRenderParams` (`src/world.rs:18`, `src/rendering.rs:6-7`,
/// cargo feature `mis` / `brdf_only`, `Cargo.toml:6-10`).
pub fn params() -> pt::RenderParams {
    let mut p = pt::default_params(); // spp 3000, MIN_DEPTH 4, MAX_DEPTH 50, t_min 1e-3
    p.spp = SAMPLE_NUM;
    p.integrator = if cfg!(feature = "brdf_only") { 1 } else { 0 };
    p
}

fn store(world: &World, rgba8: &[u8], linear_rgb: Option<&[f32]>) {
    let n = (WIDTH * HEIGHT) as usize;
    {
        let mut data = world.data.lock().unwrap(); // main.rs:59
        for i in 0..n {
            data[i] = Color { r: rgba8[4 * i], g: rgba8[4 * i + 1], b: rgba8[4 * i + 2], a: rgba8[4 * i + 3] };
        }
    }
    if let Some(lin) = linear_rgb {
        let mut lum = world.luminance_data.lock().unwrap(); // world.rs:318-319
        for i in 0..n {
            lum[i] = Vector3::new(lin[3 * i] as f64, lin[3 * i + 1] as f64, lin[3 * i + 2] as f64);
        }
    }
}

/// Everything `main.rs:43-60` does, on `devices` (one GPU: `&[0]`; several: interleaved row bands and one
/// RCCL gather of the film, `pt_multi_*`).  While it renders, `World.data` is refreshed every `preview_spp`
/// samples so that the window keeps showing the image converge (`main.rs:79-90`); the last refresh is the
/// finished film.  Blocking; call it from the render thread spawned at `main.rs:42`.
pub fn render(world: &World, devices: &[i32], preview_spp: u32) -> Result<(), pt::Error> {
    let objects = flatten(world);
    let camera = world.camera().to_pod();
    assert!(camera.width == WIDTH && camera.height == HEIGHT, "World.data is a fixed WIDTH x HEIGHT array (world.rs:55)");
    let p = params();
    if devices.len() > 1 {
        let mut multi = pt::Multi::new(devices)?;
        multi.upload(&objects)?;
        let film = multi.render(&camera, &p)?;
        store(world, &film.rgba8, Some(&film.linear_rgb));
        return Ok(());
    }
    let mut renderer = pt::Renderer::new(devices.first().copied().unwrap_or(0))?;
    renderer.upload(&objects)?;
    let film = renderer.render_progressive(&camera, &p, preview_spp, |_done, _total, rgba8| {
        store(world, rgba8, None);
        false // never stop early
    })?;
    store(world, &film.rgba8, Some(&film.linear_rgb));
    Ok(())
}

/// `World::render_pixel(x, y, rng)` (`src/world.rs:293-333`) for a list of pixels: each gets exactly the
/// samples the full render gives it.  Returns (Color, linear mean) per pixel and, like `render_pixel`, stores
/// the linear mean into `luminance_data`.
pub fn render_pixels(world: &World, pixels: &[(u32, u32)]) -> Result<Vec<(Color, Vector3)>, pt::Error> {
    let mut renderer = pt::Renderer::new(0)?;
    renderer.upload(&flatten(world))?;
    let out = renderer.render_pixels(&world.camera().to_pod(), &params(), pixels, false)?;
    let mut lum = world.luminance_data.lock().unwrap();
    Ok(pixels
        .iter()
        .enumerate()
        .map(|(i, &(x, y))| {
            let v = Vector3::new(out.linear_rgb[3 * i] as f64, out.linear_rgb[3 * i + 1] as f64, out.linear_rgb[3 * i + 2] as f64);
            lum[(y * WIDTH + x) as usize] = v;
            (Color { r: out.rgba8[4 * i], g: out.rgba8[4 * i + 1], b: out.rgba8[4 * i + 2], a: out.rgba8[4 * i + 3] }, v)
        })
        .collect())
}
