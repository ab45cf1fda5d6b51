//! bindings/ofdm_hip.rs -- Rust FFI for libofdm_hip.so (include/ofdm_hip.h).
//!
//! UNTESTED SOURCE TEXT: the authoring image has no cargo/rustc.  It shows the binding a
//! maintainer of jkelleyrtp/ofdm would add as `src/ofdm_hip.rs` so that `encode` / `decode`
//! (src/transmitter.rs:11-58, src/receiver.rs:9-96) keep their signatures while the DSP runs on an MI355X.
#![allow(non_camel_case_types, dead_code)]
use num::complex::Complex64;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct ofdm_ctx { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ofdm_fc32 { pub re: f32, pub im: f32 }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ofdm_params {
    pub n_fft: i32, pub cp_len: i32, pub modulation: i32, pub guard_bands: i32, pub ecc: i32,
    pub sync_window_reps: i32, pub sync_backoff: i32, pub cfo_mode: i32, pub sync_threshold: f32,
    pub sync_mode: i32,
    pub rx_path: i32,
    pub reserved: [i32; 5],
}

pub const OFDM_OK: c_int = 0;
pub const OFDM_FRAME_SHORT: i32 = -1; // "Input not long enough, bailing early" (src/receiver.rs:27-29)
pub const OFDM_MOD_BPSK: i32 = 1;
pub const OFDM_MOD_QPSK: i32 = 2;
pub const OFDM_MOD_QAM16: i32 = 4;
pub const OFDM_MOD_QAM64: i32 = 6;

/// The crate's own `ModulationScheme` (src/transmitter.rs:98-104) is what `encode` / `decode` keep taking.  Its `Qam` arm is
/// empty in the reference (transmitter.rs:135-136, receiver.rs:185: "Only 16 qam is implemented"); here it selects 16-QAM.
use crate::ModulationScheme;
fn bits_per_point(m: &ModulationScheme) -> i32 {
    match m {
        ModulationScheme::Bpsk => OFDM_MOD_BPSK,
        ModulationScheme::Qpsk => OFDM_MOD_QPSK,
        ModulationScheme::Qam => OFDM_MOD_QAM16,
    }
}

#[link(name = "ofdm_hip")]
extern "C" {
    pub fn ofdm_abi_version() -> c_int;
    pub fn ofdm_strerror(status: c_int) -> *const c_char;
    pub fn ofdm_device_count(count: *mut c_int) -> c_int;
    pub fn ofdm_default_params(p: *mut ofdm_params) -> c_int;
    pub fn ofdm_default_pilots(n_fft: i32, cp_len: i32, preamble: *mut f64, training: *mut f64) -> c_int;
    pub fn ofdm_stdrng_pilots(n_fft: i32, cp_len: i32, preamble: *mut f64, training: *mut f64) -> c_int;
    pub fn ofdm_rs255_encoded_len(n_bytes: i64) -> i64;
    pub fn ofdm_rs255_decoded_len(n_code: i64) -> i64;
    pub fn ofdm_rs255_encode(data: *const u8, n_bytes: i64, out: *mut u8) -> c_int;
    pub fn ofdm_rs255_decode(code: *const u8, n_code: i64, out: *mut u8, corrected: *mut i32) -> c_int;
    pub fn ofdm_tx_symbols_batch(ctx: *mut ofdm_ctx, bytes_dev: *const u8, n_bytes: i64, out_dev: *mut ofdm_fc32, n_sym: i64) -> c_int;
    pub fn ofdm_chacha_block(key8: *const u32, words12_15: *const u32, rounds: i32, out16: *mut u32) -> c_int;
    pub fn ofdm_create(p: *const ofdm_params, preamble: *const f64, training: *const f64, device: c_int,
                       stream: *mut c_void, out: *mut *mut ofdm_ctx) -> c_int;
    pub fn ofdm_destroy(ctx: *mut ofdm_ctx) -> c_int;
    pub fn ofdm_set_stream(ctx: *mut ofdm_ctx, stream: *mut c_void) -> c_int;
    pub fn ofdm_synchronize(ctx: *mut ofdm_ctx) -> c_int;
    pub fn ofdm_last_hip_error(ctx: *const ofdm_ctx) -> c_int;
    pub fn ofdm_last_dispatch(ctx: *const ofdm_ctx, buf: *mut c_char, n: usize) -> c_int;
    pub fn ofdm_set_tuning(ctx: *mut ofdm_ctx, key: *const c_char, value: i64) -> c_int;
    pub fn ofdm_get_tuning(ctx: *const ofdm_ctx, key: *const c_char, value: *mut i64) -> c_int;
    pub fn ofdm_dev_alloc(ctx: *mut ofdm_ctx, bytes: usize, dev: *mut *mut c_void) -> c_int;
    pub fn ofdm_dev_free(ctx: *mut ofdm_ctx, dev: *mut c_void) -> c_int;
    pub fn ofdm_memcpy_h2d(ctx: *mut ofdm_ctx, dev: *mut c_void, host: *const c_void, bytes: usize) -> c_int;
    pub fn ofdm_memcpy_d2h(ctx: *mut ofdm_ctx, host: *mut c_void, dev: *const c_void, bytes: usize) -> c_int;
    pub fn ofdm_memset(ctx: *mut ofdm_ctx, dev: *mut c_void, value: c_int, bytes: usize) -> c_int;
    pub fn ofdm_symbol_len(ctx: *const ofdm_ctx) -> c_int;
    pub fn ofdm_data_carriers(ctx: *const ofdm_ctx) -> c_int;
    pub fn ofdm_bytes_per_symbol(ctx: *const ofdm_ctx) -> c_int;
    pub fn ofdm_coded_len(ctx: *const ofdm_ctx, payload_bytes: i64) -> i64;
    pub fn ofdm_data_symbols(ctx: *const ofdm_ctx, payload_bytes: i64) -> i64;
    pub fn ofdm_frame_samples(ctx: *const ofdm_ctx, payload_bytes: i64) -> i64;
    pub fn ofdm_fft_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, out_dev: *mut ofdm_fc32, n_vec: i64, inverse: c_int) -> c_int;
    pub fn ofdm_ifft_cp_batch(ctx: *mut ofdm_ctx, freq_dev: *const ofdm_fc32, out_dev: *mut ofdm_fc32, n_sym: i64) -> c_int;
    pub fn ofdm_unprefix_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, out_dev: *mut ofdm_fc32, n_sym: i64) -> c_int;
    pub fn ofdm_qam_map_batch(ctx: *mut ofdm_ctx, bytes_dev: *const u8, n_bytes: i64, out_dev: *mut ofdm_fc32) -> c_int;
    pub fn ofdm_qam_demap_batch(ctx: *mut ofdm_ctx, sym_dev: *const ofdm_fc32, n_sym: i64, bytes_dev: *mut u8, idx_dev: *mut u8) -> c_int;
    pub fn ofdm_encode_block_batch(ctx: *mut ofdm_ctx, data_dev: *const ofdm_fc32, bins_dev: *mut ofdm_fc32, n_sym: i64) -> c_int;
    pub fn ofdm_normalize_batch(ctx: *mut ofdm_ctx, x_dev: *mut ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64) -> c_int;
    pub fn ofdm_hamming74_encode(ctx: *mut ofdm_ctx, in_dev: *const u8, n_bytes: i64, out_dev: *mut u8) -> c_int;
    pub fn ofdm_hamming74_decode(ctx: *mut ofdm_ctx, in_dev: *const u8, n_bytes: i64, out_dev: *mut u8, corrected_dev: *mut u32) -> c_int;
    pub fn ofdm_sc_correlate_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64,
                                   n_lags: i64, d_hat_dev: *mut i32, f_delta_dev: *mut f64, metric_dev: *mut f32) -> c_int;
    pub fn ofdm_frequency_correction_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_pairs: i64, stride: i64,
                                           right_offset: i64, f_delta_dev: *mut f64) -> c_int;
    pub fn ofdm_cfo_rotate_batch(ctx: *mut ofdm_ctx, x_dev: *mut ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64,
                                 f_delta_dev: *const f64, first_index_dev: *const i32) -> c_int;
    pub fn ofdm_estimate_channel_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_frames: i64, frame_stride: i64,
                                       frame_len: i64, offset_dev: *const i32, f_delta_dev: *const f64, hk_dev: *mut ofdm_fc32) -> c_int;
    pub fn ofdm_rx_demod_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64,
                               first_symbol: i32, syms_per_frame: i32, offset_dev: *const i32, f_delta_dev: *const f64,
                               hk_dev: *const ofdm_fc32, hk_stride: i64, out_dev: *mut u8, out_stride: i64, soft_dev: *mut ofdm_fc32) -> c_int;
    pub fn ofdm_tx_encode_batch(ctx: *mut ofdm_ctx, payload_dev: *const u8, n_frames: i64, payload_stride: i64,
                                payload_len_dev: *const i32, payload_bytes: i32, out_dev: *mut ofdm_fc32, out_stride: i64) -> c_int;
    pub fn ofdm_rx_decode_batch(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64,
                                n_lags: i64, max_symbols: i32, out_dev: *mut u8, out_stride: i64, out_len_dev: *mut i32,
                                status_dev: *mut i32, offset_dev: *mut i32, f_delta_dev: *mut f64, metric_dev: *mut f32) -> c_int;
    pub fn ofdm_xcorr_batch(ctx: *mut ofdm_ctx, a_dev: *const ofdm_fc32, n_frames: i64, a_stride: i64, a_len: i64, b_dev: *const ofdm_fc32, nb: i32, idx_max_dev: *mut i32, peak_dev: *mut f32, out_dev: *mut ofdm_fc32, out_stride: i64) -> c_int;
    pub fn ofdm_channel_batch(ctx: *mut ofdm_ctx, tx_dev: *const ofdm_fc32, n_frames: i64, tx_stride: i64, tx_len: i64, snr_db: f64, timing_error: i32, seed: u64, delay_dev: *const i32, f_delta_in_dev: *const f64, out_dev: *mut ofdm_fc32, out_stride: i64, out_len: i64, f_delta_out_dev: *mut f64) -> c_int;
    pub fn ofdm_channel_taps(taps64: *mut f64) -> c_int;
    pub fn ofdm_hbm_read_probe(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_symbols: i64, pattern: i32) -> c_int;
    pub fn ofdm_use_own_stream(ctx: *mut ofdm_ctx) -> c_int;
    pub fn ofdm_host_alloc(bytes: usize, host: *mut *mut c_void) -> c_int;
    pub fn ofdm_host_free(host: *mut c_void) -> c_int;
    pub fn ofdm_host_register(host: *mut c_void, bytes: usize) -> c_int;
    pub fn ofdm_host_unregister(host: *mut c_void) -> c_int;
    pub fn ofdm_host_is_pinned(host: *const c_void, bytes: usize) -> c_int;
    pub fn ofdm_rx_decode_host(ctx: *mut ofdm_ctx, in_host: *const ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64, n_lags: i64,
                               max_symbols: i32, out_host: *mut u8, out_stride: i64, out_len_host: *mut i32, status_host: *mut i32,
                               offset_host: *mut i32, f_delta_host: *mut f64, metric_host: *mut f32, chunk_frames: i64) -> c_int;
    pub fn ofdm_rx_demod_host(ctx: *mut ofdm_ctx, in_host: *const ofdm_fc32, n_frames: i64, frame_stride: i64, frame_len: i64,
                              first_symbol: i32, syms_per_frame: i32, out_host: *mut u8, out_stride: i64, chunk_frames: i64) -> c_int;
    pub fn ofdm_tx_encode_host(ctx: *mut ofdm_ctx, payload_host: *const u8, n_frames: i64, payload_stride: i64, payload_len_host: *const i32,
                               payload_bytes: i32, out_host: *mut ofdm_fc32, out_stride: i64, chunk_frames: i64) -> c_int;
    pub fn ofdm_sc_correlate_long(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_samples: i64, lag_lo: i64, lag_hi: i64, slice_lags: i64,
                                  d_hat: *mut i64, f_delta: *mut f64, metric: *mut f32) -> c_int;
    pub fn ofdm_rx_decode_long(ctx: *mut ofdm_ctx, in_dev: *const ofdm_fc32, n_samples: i64, lag_lo: i64, lag_hi: i64, d_hat_known: i64,
                               max_symbols: i32, out_dev: *mut u8, out_cap: i64, out_len: *mut i32, status: *mut i32, offset: *mut i64,
                               f_delta: *mut f64, metric: *mut f32) -> c_int;
    pub fn ofdm_rx_decode_long_host(ctx: *mut ofdm_ctx, in_host: *const ofdm_fc32, n_samples: i64, max_symbols: i32, out_host: *mut u8,
                                    out_cap: i64, out_len: *mut i32, status: *mut i32, offset: *mut i64, f_delta: *mut f64,
                                    metric: *mut f32) -> c_int;
    pub fn ofdm_timer_start(ctx: *mut ofdm_ctx) -> c_int;
    pub fn ofdm_timer_stop_ms(ctx: *mut ofdm_ctx, elapsed_ms: *mut f32) -> c_int;
}

/// Owning wrapper: one context per (thread, GPU).
pub struct Ctx { raw: *mut ofdm_ctx, s: usize, bytes_per_symbol: usize }

fn check(rc: c_int, what: &str) -> anyhow::Result<()> {
    if rc == OFDM_OK { Ok(()) } else {
        let msg = unsafe { std::ffi::CStr::from_ptr(ofdm_strerror(rc)) }.to_string_lossy().into_owned();
        Err(anyhow::anyhow!("{}: {}", what, msg))
    }
}

impl Ctx {
    pub fn new(guard_bands: bool, modulation: i32) -> anyhow::Result<Self> { Self::on_device(guard_bands, modulation, 0) }
    pub fn on_device(guard_bands: bool, modulation: i32, device: i32) -> anyhow::Result<Self> {
        let mut p: ofdm_params = unsafe { std::mem::zeroed() };
        check(unsafe { ofdm_default_params(&mut p) }, "ofdm_default_params")?;
        p.guard_bands = guard_bands as i32;
        p.modulation = modulation;
        let mut raw = std::ptr::null_mut();
        check(unsafe { ofdm_create(&p, std::ptr::null(), std::ptr::null(), device, std::ptr::null_mut(), &mut raw) }, "ofdm_create")?;
        let s = unsafe { ofdm_symbol_len(raw) } as usize;
        let bps = unsafe { ofdm_bytes_per_symbol(raw) } as usize;
        Ok(Ctx { raw, s, bytes_per_symbol: bps })
    }
    fn alloc(&self, bytes: usize) -> anyhow::Result<*mut c_void> {
        let mut d = std::ptr::null_mut();
        check(unsafe { ofdm_dev_alloc(self.raw, bytes, &mut d) }, "ofdm_dev_alloc")?;
        Ok(d)
    }
}
impl Drop for Ctx { fn drop(&mut self) { unsafe { ofdm_destroy(self.raw); } } }

/// One context per (thread, parameter set), created on first use and kept for the life of the thread: the reference calls `encode!` /
/// `decode!` once per frame (examples/lab3a.rs:24,34, jetson_rx.rs:86) and `ofdm_create` -- table uploads, streams, workspaces --
/// must not be paid on every call (round 3 did: 1.6 ms per call against 0.1 ms for the decode itself).
thread_local! {
    static CTX_CACHE: std::cell::RefCell<std::collections::HashMap<(bool, i32), std::rc::Rc<Ctx>>> = Default::default();
}
fn ctx_for(guard_bands: bool, modulation: i32) -> anyhow::Result<std::rc::Rc<Ctx>> {
    CTX_CACHE.with(|c| {
        let mut c = c.borrow_mut();
        if let Some(x) = c.get(&(guard_bands, modulation)) { return Ok(x.clone()); }
        let x = std::rc::Rc::new(Ctx::new(guard_bands, modulation)?);
        c.insert((guard_bands, modulation), x.clone());
        Ok(x)
    })
}

/// `ofdm::encode!(data, guard_bands, modulation)` -- the reference's exact signature (src/transmitter.rs:10-15), so the
/// `#[optargs::optfn]` attribute and every `encode!` call site stay as they are.  The reference cannot fail here; a GPU can
/// (no device, out of memory): like the reference's own `unwrap`s this panics rather than change the return type.
#[optargs::optfn]
pub fn encode(data: &[u8], guard_bands: Option<bool>, modulation: Option<crate::ModulationScheme>) -> Vec<Complex64> {
    try_encode(data, guard_bands, modulation).expect("ofdm_hip encode")
}

/// Fallible form for hosts that prefer a `Result` over the panic.  Host bytes in, host samples out: `ofdm_tx_encode_host` stages
/// through the library's pinned buffers (no device allocation per call).
pub fn try_encode(data: &[u8], guard_bands: Option<bool>, modulation: Option<crate::ModulationScheme>) -> anyhow::Result<Vec<Complex64>> {
    let ctx = ctx_for(guard_bands.unwrap_or(false), bits_per_point(&modulation.unwrap_or(ModulationScheme::Bpsk)))?;
    let n = unsafe { ofdm_frame_samples(ctx.raw, data.len() as i64) } as usize;
    let mut host = vec![ofdm_fc32 { re: 0.0, im: 0.0 }; n];
    let none = [0u8];
    let src = if data.is_empty() { none.as_ptr() } else { data.as_ptr() };
    check(unsafe { ofdm_tx_encode_host(ctx.raw, src, 1, data.len() as i64, std::ptr::null(), data.len() as i32, host.as_mut_ptr(), n as i64, 0) },
          "ofdm_tx_encode_host")?;
    Ok(host.iter().map(|c| Complex64::new(c.re as f64, c.im as f64)).collect()) // bytes_to_sig, src/utils.rs:238-254
}

/// `ofdm::decode!(samples, guard_bands, modulation)` -- the reference's exact signature (src/receiver.rs:8-13).  ONE capture of any
/// length: the 2 M-sample buffers of examples/jetson_rx.rs:15-17,84-86 are searched as a batch of overlapping slices
/// (`ofdm_rx_decode_long_host`), a frame-sized capture takes the batch path directly.
#[optargs::optfn]
pub fn decode(samples: Vec<Complex64>, guard_bands: Option<bool>, modulation: Option<crate::ModulationScheme>) -> anyhow::Result<Vec<u8>> {
    let ctx = ctx_for(guard_bands.unwrap_or(false), bits_per_point(&modulation.unwrap_or(ModulationScheme::Bpsk)))?;
    let n = samples.len();
    let fc32: Vec<ofdm_fc32> = samples.iter().map(|c| ofdm_fc32 { re: c.re as f32, im: c.im as f32 }).collect(); // sig_to_bytes
    let max_sym = (((n + ctx.s - 1) / ctx.s).saturating_sub(10)).max(1);
    let cap = (max_sym * ctx.bytes_per_symbol).max(4);
    let mut bytes = vec![0u8; cap];
    let (mut len, mut status, mut offset) = (0i32, 0i32, 0i64);
    check(unsafe { ofdm_rx_decode_long_host(ctx.raw, fc32.as_ptr(), n as i64, max_sym as i32, bytes.as_mut_ptr(), cap as i64, &mut len, &mut status,
                                            &mut offset, std::ptr::null_mut(), std::ptr::null_mut()) }, "ofdm_rx_decode_long_host")?;
    match status {
        0 => { bytes.truncate(len as usize); Ok(bytes) }
        OFDM_FRAME_SHORT => Err(anyhow::anyhow!("Input not long enough, bailing early")),
        s => Err(anyhow::anyhow!("decode failed (frame status {})", s)),
    }
}

/// Frame-index data parallelism inside one process (the C++ twin is `ofdm::ShardedContext` in include/ofdm_host.hpp):
/// one context per entry of `devices` (an ordinal may repeat), each with its own stream (`ofdm_use_own_stream`), one scoped thread per
/// context for the duration of a call, frames [r F / R, (r + 1) F / R) to context r, results written straight into the caller's
/// arrays -- no exchange between the devices.  `Ctx` is `Send` (a context may move between threads; it must not be shared).
pub struct ShardedCtx { ctxs: Vec<Ctx> }
unsafe impl Send for Ctx {}
impl ShardedCtx {
    pub fn new(devices: &[i32], guard_bands: bool, modulation: i32) -> anyhow::Result<Self> {
        let mut ctxs = Vec::new();
        for &d in devices {
            let c = Ctx::on_device(guard_bands, modulation, d)?;
            check(unsafe { ofdm_use_own_stream(c.raw) }, "ofdm_use_own_stream")?;
            ctxs.push(c);
        }
        Ok(ShardedCtx { ctxs })
    }
    /// decode_batch on host memory: `frames` holds n_frames rows of `stride` fc32 samples; returns (bytes rows of `row` bytes, len, status)
    pub fn decode_batch(&mut self, frames: &[ofdm_fc32], n_frames: usize, stride: usize, max_symbols: i32) -> anyhow::Result<(Vec<u8>, usize, Vec<i32>, Vec<i32>)> {
        let row = ((max_symbols as usize) * self.ctxs[0].bytes_per_symbol).saturating_sub(16).max(4);
        let (mut bytes, mut len, mut status) = (vec![0u8; n_frames * row], vec![0i32; n_frames], vec![0i32; n_frames]);
        let r = self.ctxs.len();
        let rcs: Vec<c_int> = std::thread::scope(|sc| {
            let mut hs = Vec::new();
            let (mut b, mut l, mut s) = (&mut bytes[..], &mut len[..], &mut status[..]);
            for (i, c) in self.ctxs.iter_mut().enumerate() {
                let (lo, hi) = (n_frames * i / r, n_frames * (i + 1) / r);
                let (bi, br) = b.split_at_mut((hi - lo) * row); b = br;
                let (li, lr) = l.split_at_mut(hi - lo); l = lr;
                let (si, sr) = s.split_at_mut(hi - lo); s = sr;
                let src = &frames[lo * stride..];
                hs.push(sc.spawn(move || unsafe {
                    ofdm_rx_decode_host(c.raw, src.as_ptr(), (hi - lo) as i64, stride as i64, stride as i64, 0, max_symbols, bi.as_mut_ptr(), row as i64,
                                        li.as_mut_ptr(), si.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), 0)
                }));
            }
            hs.into_iter().map(|h| h.join().unwrap()).collect()
        });
        for rc in rcs { check(rc, "ofdm_rx_decode_host")?; }
        Ok((bytes, row, len, status))
    }
}
