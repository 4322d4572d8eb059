//! `tinyslam::orb` over libtinyorb (include/tinyorb.h).  Same public items as the reference module:
//! `OrbConfig`, `OrbProgram::{init, write_input_image, set_threshold, extract_corners, read_corners,
//! read_descriptors}`, `CornerData`, `CornerDescriptor`.  UNVERIFIED SOURCE: the build image has no rustc; the
//! same ABI is exercised from Python by the repository's tests.
pub mod orb {
    use std::os::raw::{c_char, c_int, c_void};

    #[repr(C)]
    #[derive(Clone, Copy, Default, Debug, PartialEq, Eq)]
    pub struct CornerData {
        pub x: u32,
        pub y: u32,
        pub angle: u32,
        pub octave: u32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, PartialEq, Eq)]
    pub struct CornerDescriptor {
        pub bits: [u8; 32],
    }
    impl Default for CornerDescriptor {
        fn default() -> Self {
            Self { bits: [0; 32] }
        }
    }

    /// Stand-in for `wgpu::Extent3d` (the only wgpu type in the reference's public surface).
    #[derive(Clone, Copy, Debug)]
    pub struct Extent3d {
        pub width: u32,
        pub height: u32,
        pub depth_or_array_layers: u32,
    }

    pub struct OrbConfig {
        pub image_size: Extent3d,
        pub max_features: u32,
        pub hierarchy_depth: u32,
        pub initial_threshold: f32,
    }

    #[repr(C)]
    struct Extent3dC {
        width: u32,
        height: u32,
        depth_or_array_layers: u32,
    }
    #[repr(C)]
    struct OrbConfigC {
        image_size: Extent3dC,
        max_features: u32,
        hierarchy_depth: u32,
        initial_threshold: f32,
    }
    #[repr(C)]
    #[derive(Default)]
    struct OrbOptionsC {
        device: i32,
        max_batch: u32,
        flags: u32,    // 0 = the reference's literal algorithm
        fast_arc: u32, // 0 = FAST-12
        oob_policy: u32,          // ORB_OOB_ZERO / _CLAMP / _UMIN: textureLoad outside the level (0 = the default, CRD-6)
        sampler_weight_bits: u32, // 0 = exact bilinear weights (CRD-5); n = weights held in n fractional bits
        fp_contract: u32,         // CRD-13: mask of ORB_FP_* -- which stages' products and sums the adapter's shader compiler fuses, and its reduction order
        angle_bins: u32,          // ORB_FLAG_INTENDED only (IM-6b); 0 for the reference's algorithm
    }

    /// The points the reference's WGSL leaves to its adapter, as switches (`OrbOptions` of include/tinyorb.h).
    /// `tools/pin_oracle.py check <dump>` says which setting a given adapter follows.
    #[derive(Clone, Copy, Default)]
    pub struct AdapterBehaviour {
        pub oob_policy: u32,
        pub sampler_weight_bits: u32,
        /// CRD-13: a mask of `ORB_FP_CONTRACT_LUMINANCE | _BLUR | _ROTATION` (that stage's product-and-sum pairs as fused multiply-adds)
        /// and `ORB_FP_LAST_TERM_FIRST` (dot() and matrix * vector reduced from the last term, as Mesa lowers them); carried by the
        /// fused kernels at full speed since ABI 5
        pub fp_contract: u32,
    }
    pub const ORB_OOB_ZERO: u32 = 0;
    pub const ORB_OOB_CLAMP: u32 = 1;
    pub const ORB_OOB_UMIN: u32 = 2;
    pub const ORB_FP_CONTRACT_LUMINANCE: u32 = 1;
    pub const ORB_FP_CONTRACT_BLUR: u32 = 2;
    pub const ORB_FP_CONTRACT_ROTATION: u32 = 4;
    pub const ORB_FP_CONTRACT_ALL: u32 = 7;
    pub const ORB_FP_LAST_TERM_FIRST: u32 = 8;
    /// `OrbOptions::flags`: extract_corners spins for 50 us, then sleeps until the completion interrupt (the reference's `device.poll(Wait)`)
    pub const ORB_FLAG_SINGLE_BLOCKING_WAIT: u32 = 32;

    extern "C" {
        fn orb_program_create(cfg: *const OrbConfigC, opt: *const OrbOptionsC, out: *mut *mut c_void) -> c_int;
        fn orb_program_destroy(p: *mut c_void);
        fn orb_last_error(p: *const c_void) -> *const c_char;
        fn orb_write_input_image(p: *mut c_void, bytes: *const u8, len: usize) -> c_int;
        fn orb_write_input_image_pinned(p: *mut c_void, bytes_pinned: *const u8, len: usize) -> c_int;
        fn orb_upload_sync(p: *mut c_void) -> c_int;
        fn orb_set_threshold(p: *mut c_void, threshold: f32) -> c_int;
        fn orb_extract_corners(p: *mut c_void, corner_count: *mut u32) -> c_int;
        fn orb_read_corners(p: *mut c_void, dst: *mut CornerData, n: usize) -> c_int;
        fn orb_read_descriptors(p: *mut c_void, dst: *mut CornerDescriptor, n: usize) -> c_int;
    }

    const ORB_OK: c_int = 0;
    const ORB_ECAPACITY: c_int = 3;

    pub struct OrbProgram {
        pub config: OrbConfig,
        /// what the reference's adapter does where WGSL leaves it open (default: zeros / exact weights)
        pub adapter: AdapterBehaviour,
        handle: *mut c_void,
    }

    impl OrbProgram {
        /// The reference builds `OrbProgram { config, compute, storage }` by struct literal; the wgpu
        /// objects are gone, so construction takes the config only.
        pub fn new(config: OrbConfig) -> Self {
            Self { config, adapter: AdapterBehaviour::default(), handle: std::ptr::null_mut() }
        }

        fn check(&self, rc: c_int) {
            if rc != ORB_OK && rc != ORB_ECAPACITY {
                let msg = unsafe { std::ffi::CStr::from_ptr(orb_last_error(self.handle)) };
                panic!("tinyorb: {}", msg.to_string_lossy()); // the reference panics on every failure too
            }
        }

        pub fn init(&mut self) {
            let c = OrbConfigC {
                image_size: Extent3dC {
                    width: self.config.image_size.width,
                    height: self.config.image_size.height,
                    depth_or_array_layers: self.config.image_size.depth_or_array_layers,
                },
                max_features: self.config.max_features,
                hierarchy_depth: self.config.hierarchy_depth,
                initial_threshold: self.config.initial_threshold,
            };
            let opt = OrbOptionsC {
                oob_policy: self.adapter.oob_policy,
                sampler_weight_bits: self.adapter.sampler_weight_bits,
                fp_contract: self.adapter.fp_contract,
                ..Default::default()
            };
            let rc = unsafe { orb_program_create(&c, &opt, &mut self.handle) };
            self.check(rc);
        }

        pub fn write_input_image(&self, bytes: &[u8]) {
            self.check(unsafe { orb_write_input_image(self.handle, bytes.as_ptr(), bytes.len()) });
        }

        /// The reference's non-blocking upload (orb.rs:567-583 returns after `queue.write_texture`): `bytes` must lie in
        /// pinned host memory (`orb_host_alloc`) and stay untouched until `upload_sync()`.  One image may be written
        /// ahead of `extract_corners`, so a camera loop uploads frame k + 1 under the kernels of frame k.
        ///
        /// # Safety
        /// `bytes` must point to `len` bytes of pinned memory that outlive the upload.
        pub unsafe fn write_input_image_pinned(&self, bytes: *const u8, len: usize) {
            self.check(orb_write_input_image_pinned(self.handle, bytes, len));
        }

        pub fn upload_sync(&self) {
            self.check(unsafe { orb_upload_sync(self.handle) });
        }

        pub fn set_threshold(&self, threshold: f32) {
            self.check(unsafe { orb_set_threshold(self.handle, threshold) });
        }

        /// Returns the raw detection counter (it may exceed `max_features`, as in the reference).
        pub fn extract_corners(&self) -> u32 {
            let mut n = 0u32;
            self.check(unsafe { orb_extract_corners(self.handle, &mut n) });
            n
        }

        pub fn read_corners(&self, dst: &mut [CornerData]) {
            self.check(unsafe { orb_read_corners(self.handle, dst.as_mut_ptr(), dst.len()) });
        }

        pub fn read_descriptors(&self, dst: &mut [CornerDescriptor]) {
            self.check(unsafe { orb_read_descriptors(self.handle, dst.as_mut_ptr(), dst.len()) });
        }
    }

    impl Drop for OrbProgram {
        fn drop(&mut self) {
            if !self.handle.is_null() {
                unsafe { orb_program_destroy(self.handle) }
            }
        }
    }

    // One program = one device + its streams; calls on one program must be serialised by the caller.
    unsafe impl Send for OrbProgram {}

    // ------------------------------------------------------------------------------------------------------------
    // Batched, multi-GPU entry (include/tinyorb.h "one node, several GPUs"; not in the reference, which drives one
    // wgpu device).  The same calls, in the same order, are exercised from C by examples/node_batch.c, which the
    // repository's GPU tests compile with gcc and run -- that C program is the verified twin of this block.
    // ------------------------------------------------------------------------------------------------------------
    extern "C" {
        fn orb_node_create(devices: *const c_int, n: c_int, cfg: *const OrbConfigC, opt: *const OrbOptionsC,
                           out: *mut *mut c_void) -> c_int;
        fn orb_node_destroy(node: *mut c_void);
        fn orb_node_last_error(node: *const c_void) -> *const c_char;
        fn orb_node_device_count(node: *const c_void) -> c_int;
        fn orb_node_extract_batch_host(node: *mut c_void, frames: *const u8, n_frames: u32) -> c_int;
        fn orb_node_extract_batch(node: *mut c_void, frames_dev: *const *const u8, n_frames: u32) -> c_int;
        fn orb_node_collate(node: *mut c_void, counts: *mut u32, offsets: *mut u64, corners_dev: *mut *mut c_void,
                            descriptors_dev: *mut *mut c_void) -> c_int;
        fn orb_node_collate_begin(node: *mut c_void) -> c_int;
        fn orb_node_collate_end(node: *mut c_void, counts: *mut u32, offsets: *mut u64, corners_dev: *mut *mut c_void,
                                descriptors_dev: *mut *mut c_void) -> c_int;
        fn orb_node_pending(node: *const c_void) -> c_int;
        fn orb_node_set_results(node: *mut c_void, where_: c_int) -> c_int;
        fn orb_node_shard_result(node: *mut c_void, rank: c_int, n_frames: *mut u32, n_records: *mut u64, corners_dev: *mut *mut c_void,
                                 descriptors_dev: *mut *mut c_void) -> c_int;
        fn orb_node_exchange_backend(node: *const c_void) -> *const c_char;
        fn orb_node_rccl_pairs(node: *const c_void) -> u64;
        fn orb_node_read_collated(node: *mut c_void, corners: *mut CornerData, descriptors: *mut CornerDescriptor,
                                  capacity: usize) -> c_int;
    }

    /// Results of one job: the stored records of all frames back to back, frame `f` owns `offsets[f]..offsets[f+1]`.
    pub struct BatchResult {
        pub counts: Vec<u32>,   // raw per-frame counters (may exceed max_features, like `extract_corners`)
        pub offsets: Vec<u64>,  // n_frames + 1
        pub corners: Vec<CornerData>,
        pub descriptors: Vec<CornerDescriptor>,
    }

    /// One process, several GPUs: frames are sharded in contiguous ranges over `devices`, collated on the first.
    pub struct OrbNode {
        handle: *mut c_void,
        frame_bytes: usize,
    }

    impl OrbNode {
        /// The constructor as it was before `flags` (round 2's signature): the reference's algorithm on RGBA frames.
        pub fn with_defaults(devices: &[i32], config: &OrbConfig, max_batch: u32) -> Self {
            Self::new(devices, config, max_batch, 0)
        }

        /// `max_batch` = the largest shard one device may get (frames per job / devices, rounded up).
        /// `flags`: ORB_FLAG_* of include/tinyorb.h; with ORB_FLAG_INPUT_Y8 (16) frames are one byte per pixel.
        /// (Round 3 added `flags`: callers of the three-argument form use `with_defaults`.)
        pub fn new(devices: &[i32], config: &OrbConfig, max_batch: u32, flags: u32) -> Self {
            let c = OrbConfigC {
                image_size: Extent3dC {
                    width: config.image_size.width,
                    height: config.image_size.height,
                    depth_or_array_layers: config.image_size.depth_or_array_layers,
                },
                max_features: config.max_features,
                hierarchy_depth: config.hierarchy_depth,
                initial_threshold: config.initial_threshold,
            };
            let opt = OrbOptionsC { max_batch, flags, ..Default::default() };
            let mut handle = std::ptr::null_mut();
            let rc = unsafe { orb_node_create(devices.as_ptr(), devices.len() as c_int, &c, &opt, &mut handle) };
            if rc != ORB_OK {
                let msg = unsafe { std::ffi::CStr::from_ptr(orb_node_last_error(std::ptr::null())) };
                panic!("tinyorb: {}", msg.to_string_lossy());
            }
            const ORB_FLAG_INPUT_Y8: u32 = 16;
            let bytes_per_pixel = if flags & ORB_FLAG_INPUT_Y8 != 0 { 1 } else { 4 };  // the node slices the host array the same way
            let frame_bytes = config.image_size.width as usize * config.image_size.height as usize * bytes_per_pixel;
            Self { handle, frame_bytes }
        }

        fn check(&self, rc: c_int) {
            if rc != ORB_OK {
                let msg = unsafe { std::ffi::CStr::from_ptr(orb_node_last_error(self.handle)) };
                panic!("tinyorb: {}", msg.to_string_lossy());
            }
        }

        pub fn device_count(&self) -> usize {
            unsafe { orb_node_device_count(self.handle) as usize }
        }

        /// `frames`: n tightly packed RGBA8 frames in host memory.  Shards, extracts on every device, collates.
        pub fn extract(&self, frames: &[u8]) -> BatchResult {
            let n = (frames.len() / self.frame_bytes) as u32;
            assert_eq!(frames.len(), n as usize * self.frame_bytes);
            self.check(unsafe { orb_node_extract_batch_host(self.handle, frames.as_ptr(), n) });
            self.collate(n)
        }

        /// The same for shards that are already resident: `frames_dev[r]` points to rank r's frames on ITS device.
        pub fn extract_device(&self, frames_dev: &[*const u8], n_frames: u32) -> BatchResult {
            assert_eq!(frames_dev.len(), self.device_count());
            self.check(unsafe { orb_node_extract_batch(self.handle, frames_dev.as_ptr(), n_frames) });
            self.collate(n_frames)
        }

        /// Stage 1 of a streamed job (up to two may be outstanding): the kernels of every shard, asynchronously.
        pub fn submit(&self, frames: &[u8]) -> u32 {
            let n = (frames.len() / self.frame_bytes) as u32;
            assert_eq!(frames.len(), n as usize * self.frame_bytes);
            self.check(unsafe { orb_node_extract_batch_host(self.handle, frames.as_ptr(), n) });
            n
        }

        /// Stage 2 of the oldest job that has not begun it: enqueue its exchange (overlaps the kernels of the job
        /// submitted after it).
        pub fn collate_begin(&self) {
            self.check(unsafe { orb_node_collate_begin(self.handle) });
        }

        /// Stage 3 of the oldest job (`n` = its frames): blocks until it is collated, copies the records to the host.
        ///     let n0 = node.submit(a);  let n1 = node.submit(b);  node.collate_begin();  let ra = node.finish(n0); ...
        pub fn finish(&self, n: u32) -> BatchResult {
            self.collate(n)
        }

        pub fn pending(&self) -> usize {
            unsafe { orb_node_pending(self.handle) as usize }
        }

        /// Results collated on the first device (`false`, the default) or left packed on the device that computed them
        /// (`true`: no exchange; `shard_result(rank)` hands out that rank's device buffers).  Only with no job outstanding.
        pub fn set_results_sharded(&self, sharded: bool) {
            let rc = unsafe { orb_node_set_results(self.handle, if sharded { 1 } else { 0 }) };
            assert!(rc == 0, "tinyorb: orb_node_set_results failed");
        }

        /// (frames, records, device address of the corners, of the descriptors) of `rank` for the job ended last.
        pub fn shard_result(&self, rank: i32) -> (u32, u64, *mut c_void, *mut c_void) {
            let (mut nf, mut nr) = (0u32, 0u64);
            let (mut c, mut d) = (std::ptr::null_mut(), std::ptr::null_mut());
            let rc = unsafe { orb_node_shard_result(self.handle, rank, &mut nf, &mut nr, &mut c, &mut d) };
            assert!(rc == 0, "tinyorb: orb_node_shard_result failed");
            (nf, nr, c, d)
        }

        /// How the records of ranks >= 1 reach the first device: "rccl", "rccl-self", "copies" or "none".
        pub fn exchange_backend(&self) -> String {
            unsafe { std::ffi::CStr::from_ptr(orb_node_exchange_backend(self.handle)).to_string_lossy().into_owned() }
        }

        /// ncclSend + ncclRecv pairs this node has enqueued so far.
        pub fn rccl_pairs(&self) -> u64 {
            unsafe { orb_node_rccl_pairs(self.handle) }
        }

        fn collate(&self, n: u32) -> BatchResult {
            let mut counts = vec![0u32; n as usize];
            let mut offsets = vec![0u64; n as usize + 1];
            self.check(unsafe {
                orb_node_collate_end(self.handle, counts.as_mut_ptr(), offsets.as_mut_ptr(), std::ptr::null_mut(),
                                     std::ptr::null_mut())
            });
            let total = offsets[n as usize] as usize;
            let mut corners = vec![CornerData::default(); total];
            let mut descriptors = vec![CornerDescriptor::default(); total];
            self.check(unsafe {
                orb_node_read_collated(self.handle, corners.as_mut_ptr(), descriptors.as_mut_ptr(), total)
            });
            BatchResult { counts, offsets, corners, descriptors }
        }
    }

    impl Drop for OrbNode {
        fn drop(&mut self) {
            if !self.handle.is_null() {
                unsafe { orb_node_destroy(self.handle) }
            }
        }
    }
    unsafe impl Send for OrbNode {}
}
