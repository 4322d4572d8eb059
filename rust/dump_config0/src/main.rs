//! Drives the REFERENCE's own `tinyslam::orb::OrbProgram` on one 640x480 RGBA frame (BASELINE.json configs[0]) and writes
//! what it returns -- the raw counter, the corners, the descriptors -- as three `.npy` files next to a small text record
//! of the adapter.  `tools/pin_oracle.py check <dir>` compares them with this repository's CPU restatement under every
//! setting of the two implementation-defined switches (out-of-level loads, sampler weight precision) and says which one
//! the adapter follows; the loader tests (`tests/test_oracle.py::test_reference_dump_pins_the_oracle`,
//! `tests/test_gpu_round4.py::test_reference_dump_on_gpu`) then hold the GPU path to it.
//!
//! UNVERIFIED SOURCE: written against the reference's text (src/orb.rs:40-51 the structs, :107 `init`, :567
//! `write_input_image`, :469 `extract_corners`, :559 / :563 `read_corners` / `read_descriptors`) without a Rust toolchain
//! at hand.  The one call that text does not show is how a `tiny_wgpu::Compute` is made (the reference constructs its
//! program elsewhere); tiny_wgpu 0.1.10 documents `Compute::new(features, limits).await`.
use std::{env, fs, io::Write, path::Path};

use tiny_wgpu::{Compute, Storage};
use tinyslam::orb::{CornerData, CornerDescriptor, OrbConfig, OrbProgram};

const W: u32 = 640;
const H: u32 = 480;
const MAX_FEATURES: u32 = 8192;
const DEPTH: u32 = 2;
const THRESHOLD: f32 = 20.0 / 255.0;

/// NumPy `.npy` version 1.0: magic, version, little-endian u16 header length, a Python dict literal padded with spaces to
/// a multiple of 64 bytes and ended by a newline, then the raw little-endian data in C order.
fn write_npy_u32(path: &Path, shape: &[usize], data: &[u32]) -> std::io::Result<()> {
    let dims = match shape.len() {
        0 => String::from("()"),
        1 => format!("({},)", shape[0]),
        _ => format!("({})", shape.iter().map(|d| d.to_string()).collect::<Vec<_>>().join(", ")),
    };
    let mut header = format!("{{'descr': '<u4', 'fortran_order': False, 'shape': {}, }}", dims);
    while (10 + header.len() + 1) % 64 != 0 {
        header.push(' ');
    }
    header.push('\n');
    let mut f = fs::File::create(path)?;
    f.write_all(b"\x93NUMPY\x01\x00")?;
    f.write_all(&(header.len() as u16).to_le_bytes())?;
    f.write_all(header.as_bytes())?;
    for v in data {
        f.write_all(&v.to_le_bytes())?;
    }
    Ok(())
}

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 3 && args.len() != 5 {
        eprintln!("usage: dump_config0 <frame.rgba (640x480x4 bytes, tools/pin_oracle.py frame)> <output directory> [seed flags]");
        std::process::exit(2);
    }
    // which synthetic frame this is (recorded for tools/pin_oracle.py check; default: configs[0]'s, seed 1 flags 7)
    let seed: u32 = if args.len() == 5 { args[3].parse().expect("seed") } else { 1 };
    let flags: u32 = if args.len() == 5 { args[4].parse().expect("flags") } else { 7 };
    let rgba = fs::read(&args[1]).expect("cannot read the frame");
    assert_eq!(rgba.len(), (W * H * 4) as usize, "the frame must be 640 x 480 RGBA8, tightly packed");
    let out = Path::new(&args[2]);
    fs::create_dir_all(out).expect("cannot create the output directory");

    // R16Float render targets + read/write storage textures are what tiny_wgpu asks its adapter for (README: "Enable
    // read/write storage textures", "Increase default limits for push constants and number of bindings").
    let features = wgpu::Features::TEXTURE_ADAPTER_SPECIFIC_FORMAT_FEATURES | wgpu::Features::PUSH_CONSTANTS;
    let limits = wgpu::Limits { max_push_constant_size: 4, ..wgpu::Limits::default() };
    let compute: Compute = pollster::block_on(Compute::new(features, limits));
    let info = compute.adapter.get_info();

    // orb.rs:47-51: all fields are `pub`, there is no constructor
    let mut program = OrbProgram {
        config: OrbConfig {
            image_size: wgpu::Extent3d { width: W, height: H, depth_or_array_layers: 1 },
            max_features: MAX_FEATURES,
            hierarchy_depth: DEPTH,
            initial_threshold: THRESHOLD,
        },
        compute,
        storage: Storage::default(),
    };
    program.init(); // orb.rs:107
    program.write_input_image(&rgba); // orb.rs:567
    let total = program.extract_corners(); // orb.rs:469: the RAW counter (may exceed max_features)
    let stored = total.min(MAX_FEATURES) as usize;

    // CornerData / CornerDescriptor are #[repr(C)] Pod with private fields (orb.rs:10-23): read them as bytes
    let mut corners: Vec<CornerData> = vec![bytemuck::Zeroable::zeroed(); MAX_FEATURES as usize];
    let mut descriptors: Vec<CornerDescriptor> = vec![bytemuck::Zeroable::zeroed(); MAX_FEATURES as usize];
    program.read_corners(&mut corners); // orb.rs:559
    program.read_descriptors(&mut descriptors); // orb.rs:563
    let c_words: &[u32] = bytemuck::cast_slice(&corners[..stored]); // x, y, angle, octave
    let d_words: &[u32] = bytemuck::cast_slice(&descriptors[..stored]); // 8 little-endian words, brief.wgsl:15

    write_npy_u32(&out.join("total.npy"), &[], &[total]).unwrap();
    write_npy_u32(&out.join("corners.npy"), &[stored, 4], c_words).unwrap();
    write_npy_u32(&out.join("descriptors.npy"), &[stored, 8], d_words).unwrap();
    write_npy_u32(&out.join("params.npy"), &[6], &[W, H, DEPTH, seed, flags, MAX_FEATURES]).unwrap();
    fs::write(
        out.join("adapter.txt"),
        format!("{:?}\nbackend: {:?}\ndriver: {} {}\nthreshold: {}\n", info.name, info.backend, info.driver, info.driver_info, THRESHOLD),
    )
    .unwrap();
    println!("{} corners detected, {} stored -> {}", total, stored, out.display());
}
