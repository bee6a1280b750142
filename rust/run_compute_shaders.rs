//! Replacement body of the reference's `run_compute_shaders` (src/main.rs:376-688) and the
//! readback of `run_headless` (src/main.rs:772-864) on top of `ffi.rs`.
//!
//! SOURCE ONLY (no rustc/cargo in the build image).  `create_scene`,
//! `create_morton_uniforms`, `constants.rs` and `types.rs` of the reference are used
//! unchanged; `hip` stands for any binding that provides hipMalloc/hipMemcpy/hipStream.
use crate::constants::*;
use crate::ffi as mrs;

pub struct Buffers {
    pub vertices: hip::DevicePtr,
    pub indices: hip::DevicePtr,
    pub indices_2: hip::DevicePtr,
    pub morton_uniforms: hip::DevicePtr,
    pub morton_codes: hip::DevicePtr,
    pub morton_codes_2: hip::DevicePtr,
    pub plan: mrs::Plan,          // owns what `histogram` and `prefix_sums` were (src/types.rs:73,76)
    pub result_in_2: bool,
}

pub fn run_compute_shaders(ctx: &mrs::Context, stream: hip::Stream) -> mrs::Result<Buffers> {
    let (vertices, triangles, generator, _v, _t) = create_scene();          // src/main.rs:114-173
    let uniforms = create_morton_uniforms(&generator);                       // src/main.rs:177-198
    let n = NUM_TRIANGLES as u64;
    // create_all_buffers (src/main.rs:31-85): encase-serialised bytes go up as they are
    let vertices_b = hip::upload(&encase_bytes(&vertices));                  // stride 48
    let indices_b = hip::upload(&encase_bytes(&triangles));                  // stride 16
    let uniforms_b = hip::upload(&encase_bytes(&uniforms));                  // 51248 bytes
    let indices_2_b = hip::alloc(16 * n as usize);
    let codes_b = hip::alloc(8 * n as usize);
    let codes_2_b = hip::alloc(8 * n as usize);
    let plan = mrs::Plan::new(ctx, mrs::PlanDesc {
        n, key_words: 2, payload_bytes: 16, digit_bits: BITS_PER_PASS as i32, begin_bit: 0,
        end_bit: (BITS_PER_PASS * NUM_PASSES) as i32, algo: mrs::ALGO_ONESWEEP, tile_items: 0,
    })?;
    // "Morton Coding" (src/main.rs:581-592)
    ctx.morton_encode_lut(uniforms_b.ptr(), vertices_b.ptr(), NUM_VERTICES as u64, indices_b.ptr(), n,
                          codes_b.ptr() as *mut u32, stream.raw())?;
    // histogram / prefix sums / indexing x NUM_PASSES with ping-pong (src/main.rs:594-665)
    let result_in_2 = plan.sort_pairs(n, codes_b.ptr(), indices_b.ptr(), codes_2_b.ptr(), indices_2_b.ptr(),
                                      stream.raw())?;
    Ok(Buffers { vertices: vertices_b, indices: indices_b, indices_2: indices_2_b, morton_uniforms: uniforms_b,
                 morton_codes: codes_b, morton_codes_2: codes_2_b, plan, result_in_2 })
}

pub fn run_headless() -> mrs::Result<()> {
    let ctx = mrs::Context::new(0)?;
    let stream = hip::Stream::new();
    let buffers = run_compute_shaders(&ctx, stream)?;
    buffers.plan.check(stream.raw())?;                                        // synchronises
    let out = if buffers.result_in_2 { &buffers.morton_codes_2 } else { &buffers.morton_codes };
    let codes: Vec<u64> = hip::download(out, NUM_TRIANGLES as usize);
    let mut times_unsorted = 0u32;                                            // src/main.rs:831-864
    for w in codes.windows(2) {
        if select_bits(w[0], NUM_PASSES) > select_bits(w[1], NUM_PASSES) { times_unsorted += 1; }
    }
    if times_unsorted != 0 { log::error!("Not sorted!!! {}", times_unsorted); } else { println!("Sorted !!!"); }
    Ok(())
}

/// What `State` (src/state.rs:14-22) keeps across frames once the compute path is libmrs: the
/// context, one stream, the buffer set and the plan.  `State::new` (src/state.rs:24-87) creates
/// them once; `State::render` (src/state.rs:109-148) then re-uses them, where the reference calls
/// `run_compute_shaders` and with it re-creates every buffer, pipeline and bind group per frame.
pub struct ComputeState {
    pub ctx: mrs::Context,
    pub stream: hip::Stream,
    pub buffers: Buffers,
    host_uniforms: Vec<u8>,   // this frame's uploads, alive while the stream reads them
    host_vertices: Vec<u8>,
    host_triangles: Vec<u8>,
}

impl ComputeState {
    pub fn new() -> mrs::Result<Self> {
        let ctx = mrs::Context::new(0)?;
        let stream = hip::Stream::new();
        let n = NUM_TRIANGLES as u64;
        let plan = mrs::Plan::new(&ctx, mrs::PlanDesc {
            n, key_words: 2, payload_bytes: 16, digit_bits: BITS_PER_PASS as i32, begin_bit: 0,
            end_bit: (BITS_PER_PASS * NUM_PASSES) as i32, algo: mrs::ALGO_ONESWEEP, tile_items: 0,
        })?;
        let buffers = Buffers {
            vertices: hip::alloc(48 * NUM_VERTICES as usize), indices: hip::alloc(16 * n as usize),
            indices_2: hip::alloc(16 * n as usize), morton_uniforms: hip::alloc(mrs::MORTON_UNIFORMS_BYTES),
            morton_codes: hip::alloc(8 * n as usize), morton_codes_2: hip::alloc(8 * n as usize),
            plan, result_in_2: false,
        };
        Ok(Self { ctx, stream, buffers, host_uniforms: vec![], host_vertices: vec![], host_triangles: vec![] })
    }

    /// The compute part of `State::render`: no allocation, no plan creation, no host synchronisation
    /// beyond waiting for the previous frame's uploads to have left the host arrays.
    pub fn render(&mut self) -> mrs::Result<()> {
        self.stream.synchronize();
        let (vertices, triangles, generator, _v, _t) = create_scene();
        self.host_uniforms = encase_bytes(&create_morton_uniforms(&generator));
        self.host_vertices = encase_bytes(&vertices);
        self.host_triangles = encase_bytes(&triangles);
        let b = &mut self.buffers;
        hip::upload_async(&b.morton_uniforms, &self.host_uniforms, &self.stream);
        hip::upload_async(&b.vertices, &self.host_vertices, &self.stream);
        hip::upload_async(&b.indices, &self.host_triangles, &self.stream);
        let n = NUM_TRIANGLES as u64;
        self.ctx.morton_encode_lut(b.morton_uniforms.ptr(), b.vertices.ptr(), NUM_VERTICES as u64, b.indices.ptr(), n,
                                   b.morton_codes.ptr() as *mut u32, self.stream.raw())?;
        b.result_in_2 = b.plan.sort_pairs(n, b.morton_codes.ptr(), b.indices.ptr(), b.morton_codes_2.ptr(),
                                          b.indices_2.ptr(), self.stream.raw())?;
        Ok(())
    }
}
