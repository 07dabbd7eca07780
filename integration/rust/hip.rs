//! `engine/src/net/hip.rs` -- binding of the MI355X leaf evaluator (`libcattus_hip.so`, `include/cattus_hip.h`).
//!
//! The evaluator takes a leaf's bitboard planes (what `position_to_planes` returns, as u64 words) and gives back the raw
//! policy logits (non-finite ones already replaced by `f32::MIN`, as `NNetwork::run_net` does) and the value.  It
//! replaces the closure body of `Batcher::apply` in `NNetwork::evaluate_impl` (`net/mod.rs:94-98`): batching across the
//! calling threads, `planes_to_tensor`, `Model::run` and the scrub all happen behind `cattus_hip_apply`.
//!
//! Drop this file next to `model.rs`, apply `cattus_hip.patch`, build with `--features hip` and
//! `CATTUS_HIP_LIB_DIR=<repo>/cattus_amd`.

use std::ffi::{c_char, c_void, CStr};
use std::path::Path;

/// `cattus_dtype` of include/cattus_hip.h
#[derive(Clone, Copy, Debug, PartialEq, Eq, serde::Deserialize)]
#[serde(rename_all = "lowercase")]
pub enum HipDtype {
    /// exact f32 MFMA, bit-identical to the CPU oracle's summation order
    F32 = 0,
    /// bf16 operands, f32 accumulation (throughput mode, 8 significant bits)
    Bf16 = 1,
    /// split precision (pairs of f16 values): inside the cross-runtime tolerance of `training/tests/test_net_output.py`
    F16x2 = 2,
    /// single-term f16 operands, f32 accumulation (throughput mode, 11 significant bits)
    F16 = 3,
}

#[repr(C)]
struct CattusEvalConfig {
    struct_size: u32,
    device: i32,
    max_batch: u32,
    plane_words: u32,
    dtype: u32,
    flush_us: u32,
    /// cattus_tower_form: 0 = auto (Winograd form of the f16x2 tower for batch_size >= 192), 1 = direct, 2 = Winograd
    tower_form: u32,
}

#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct CattusStats {
    /// == metric `model.activation_count` (net/mod.rs:68)
    pub batches: u64,
    pub positions: u64,
    pub full_batches: u64,
    /// == metric `model.run_duration` (EMA 0.99, util/metric.rs:16-19)
    pub run_seconds_ema: f64,
    pub run_seconds_total: f64,
    /// f16x2 / f16 towers: activation values clamped at the f16 range since creation; > 0 means this network needs dtype f32
    pub saturated: u64,
}

#[repr(C)]
struct CattusEval {
    _private: [u8; 0],
}

extern "C" {
    fn cattus_hip_create(weights: *const c_void, nbytes: usize, cfg: *const CattusEvalConfig, out: *mut *mut CattusEval) -> i32;
    fn cattus_hip_destroy(e: *mut CattusEval);
    fn cattus_hip_apply(e: *mut CattusEval, planes: *const u64, n: u32, policy: *mut f32, value: *mut f32) -> i32;
    fn cattus_hip_eval(e: *mut CattusEval, planes: *const u64, n: u32, policy: *mut f32, value: *mut f32) -> i32;
    fn cattus_hip_flush(e: *mut CattusEval) -> i32;
    fn cattus_hip_stats(e: *mut CattusEval, out: *mut CattusStats) -> i32;
    fn cattus_hip_last_error() -> *const c_char;
}

fn check(rc: i32) {
    if rc != 0 {
        // the reference panics on backend errors (model.rs:100-125,157,173,181,194)
        let msg = unsafe { CStr::from_ptr(cattus_hip_last_error()) }.to_string_lossy().into_owned();
        panic!("cattus_hip: {msg} ({rc})");
    }
}

pub struct HipModel {
    h: *mut CattusEval,
    moves: usize,
    words_per_leaf: usize,
}
// every entry point of the library is thread-safe
unsafe impl Send for HipModel {}
unsafe impl Sync for HipModel {}

impl HipModel {
    /// `path`: the weight blob `cattus_amd.weights.blob_from_state_dict` writes from `model.pt` (`model.cattus`).
    /// `planes` = planes per position, `plane_words` = `Bitboard::PLANE_WORDS`, `moves` = `Game::MOVES_NUM`.
    pub fn new(path: impl AsRef<Path>, batch_size: usize, planes: usize, plane_words: usize, moves: usize, device: i32, dtype: HipDtype) -> Self {
        let blob = std::fs::read(path.as_ref()).unwrap();
        let cfg = CattusEvalConfig {
            struct_size: std::mem::size_of::<CattusEvalConfig>() as u32,
            device,
            max_batch: batch_size.max(1) as u32,
            plane_words: plane_words as u32,
            dtype: dtype as u32,
            // partial batches run after 200 us (the reference's Batcher waits 20 ms: net/mod.rs:96)
            flush_us: 200,
            tower_form: 0,
        };
        let mut h = std::ptr::null_mut();
        check(unsafe { cattus_hip_create(blob.as_ptr().cast(), blob.len(), &cfg, &mut h) });
        Self { h, moves, words_per_leaf: planes * plane_words }
    }

    /// One leaf through the leaf-batching server: a search thread blocks here, and the leaves of all threads blocked at
    /// the same time share a batch (`Batcher::apply`, util/batch.rs:49-177).
    pub fn evaluate_planes(&self, words: &[u64]) -> (Vec<f32>, f32) {
        assert_eq!(words.len(), self.words_per_leaf);
        let mut value = 0f32;
        let mut policy = vec![0f32; self.moves];
        check(unsafe { cattus_hip_apply(self.h, words.as_ptr(), 1, policy.as_mut_ptr(), &mut value) });
        (policy, value)
    }

    /// A caller-assembled batch (1 <= n <= batch_size), blocking: `planes_to_tensor` + `run_net` for n leaves.
    pub fn evaluate_batch(&self, words: &[u64]) -> Vec<(Vec<f32>, f32)> {
        let n = words.len() / self.words_per_leaf;
        let mut values = vec![0f32; n];
        let mut policy = vec![0f32; n * self.moves];
        check(unsafe { cattus_hip_eval(self.h, words.as_ptr(), n as u32, policy.as_mut_ptr(), values.as_mut_ptr()) });
        policy.chunks(self.moves).map(|p| p.to_vec()).zip(values).collect()
    }

    pub fn flush(&self) {
        check(unsafe { cattus_hip_flush(self.h) });
    }

    pub fn stats(&self) -> CattusStats {
        let mut s = CattusStats::default();
        check(unsafe { cattus_hip_stats(self.h, &mut s) });
        s
    }
}

impl Drop for HipModel {
    fn drop(&mut self) {
        unsafe { cattus_hip_destroy(self.h) }
    }
}
