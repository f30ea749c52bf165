//! `src/asr/hip.rs` -- SOURCE ONLY (the build image has no Rust toolchain).
//! A native `AsrEngine` over libspeaksense_hip.so that replaces `WhisperAsr` (/root/reference/src/asr/whisper.rs)
//! keeping its parameter mapping (whisper.rs:131-173, 60-71) and post-processing (whisper.rs:41-43, 77-128, 175-201).
//!
//! Wiring (three one-line edits, the ones SURVEY.md section 8b names):
//!   src/asr/mod.rs:4                         `use whisper_rs::WhisperState;`  ->  `use crate::asr::hip::WhisperState;`   (+ `pub mod hip;`)
//!   src/schedule/processors/transcribe.rs:8  the same import; `:23,27` `Arc<WhisperAsr>` -> `Arc<HipAsr>`
//!   src/main.rs:38-39                        `WhisperAsr::new(path)` -> `HipAsr::new(path)`
//! `WhisperState<'a>` below keeps the NAME and lifetime parameter of whisper_rs::WhisperState, so the trait in mod.rs:58-73 and every
//! `Arc<Mutex<Box<WhisperState<'static>>>>` in the gRPC handler (grpc/handlers/asr.rs:20-22,164) compile unchanged.
use crate::asr::{AsrEngine, AsrParams, TranscribeResult, TranscribeSegment};
use anyhow::{anyhow, Result};
use std::ffi::{CStr, CString};
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int};
use std::sync::{Arc, Mutex};

#[repr(C)] pub struct ss_engine { _p: [u8; 0] }
#[repr(C)] pub struct ss_session { _p: [u8; 0] }
#[repr(C)] pub struct ss_ticket { _p: [u8; 0] }
#[repr(C)] #[derive(Default)]
pub struct ss_engine_opts { pub device: i32, pub dtype: i32, pub max_batch: i32, pub max_decoders: i32, pub batch_wait_us: i32, pub n_lanes: i32, pub compat: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct ss_params {
    pub best_of: i32, pub temperature: f32, pub temperature_inc: f32, pub entropy_thold: f32, pub logprob_thold: f32,
    pub max_initial_ts: f32, pub length_penalty: f32, pub no_context: i32, pub single_segment: i32, pub no_timestamps: i32,
    pub suppress_blank: i32, pub tdrz_enable: i32, pub print_special: i32, pub max_tokens: i32, pub audio_ctx: i32,
    pub translate: i32, pub fixed_steps: i32, pub language: [u8; 8],
    pub n_max_text_ctx: i32, pub offset_ms: i32, pub duration_ms: i32, pub detect_language: i32,
    pub prompt_tokens: *const i32, pub prompt_n_tokens: i32, pub token_timestamps: i32, pub initial_prompt: *const c_char,
    pub thold_pt: f32, pub thold_ptsum: f32,
    pub suppress_non_speech_tokens: i32, pub max_len: i32, pub split_on_word: i32,
}   // layout: tests/golden/abi_layout.txt (offsets checked against the C header by tests/test_host_cpu.py)
unsafe impl Send for ss_params {}

#[link(name = "speaksense_hip")]
extern "C" {
    fn ss_abi_version() -> i32;
    fn ss_sizeof_params() -> i32;
    fn ss_sizeof_engine_opts() -> i32;
    fn ss_default_params(p: *mut ss_params);
    fn ss_last_error() -> *const c_char;
    fn ss_engine_create(path: *const c_char, opts: *const ss_engine_opts, out: *mut *mut ss_engine) -> c_int;
    fn ss_engine_free(e: *mut ss_engine);
    fn ss_session_create(e: *mut ss_engine) -> *mut ss_session;
    fn ss_session_free(s: *mut ss_session);
    fn ss_submit(s: *mut ss_session, pcm: *const f32, n: i32, p: *const ss_params, out: *mut *mut ss_ticket) -> c_int;
    fn ss_wait(t: *mut ss_ticket) -> c_int;
    fn ss_result_n_segments(s: *const ss_session) -> i32;
    fn ss_result_segment_text(s: *const ss_session, i: i32) -> *const c_char;
    fn ss_result_segment_t0(s: *const ss_session, i: i32) -> i64;
    fn ss_result_segment_t1(s: *const ss_session, i: i32) -> i64;
    fn ss_result_segment_speaker_turn_next(s: *const ss_session, i: i32) -> i32;
}

fn last_error() -> String { unsafe { CStr::from_ptr(ss_last_error()).to_string_lossy().into_owned() } }

/// Stands where `whisper_rs::WhisperState<'a>` stood (the trait leaks that type, mod.rs:4,60,64): a session of the HIP engine.
pub struct WhisperState<'a> { raw: *mut ss_session, _ctx: PhantomData<&'a ()> }
unsafe impl<'a> Send for WhisperState<'a> {}
impl<'a> Drop for WhisperState<'a> { fn drop(&mut self) { unsafe { ss_session_free(self.raw) } } }
pub type HipSession = WhisperState<'static>;

struct EnginePtr(*mut ss_engine);
unsafe impl Send for EnginePtr {}
unsafe impl Sync for EnginePtr {}
impl Drop for EnginePtr { fn drop(&mut self) { unsafe { ss_engine_free(self.0) } } }

const PROMOTIONAL_TEXT: [&str; 14] = [
    "请不吝点赞", "請不吝點贊", "點贊", "訂閱", "订阅", "打赏", "打賞", "打賞支持明鏡與點點欄目", "打赏支持明镜与点点栏目",
    "並且按下小鈴鐺才能收到最新消息哦!", "請按讚、訂閱、分享!", "明镜需要您的支持 欢迎收看订阅明镜",
    "請按讚,訂閱,分享,打開小鈴鐺,並且按下小鈴鐺才能收到最新消息謝謝觀看",
    "請按讚,訂閱,分享,打開小鈴鐺,並且按下小鈴鐺才能收到最新消息哦!",
];

pub struct HipAsr { engine: Arc<EnginePtr> }

impl HipAsr {
    pub fn new(model_path: String) -> Result<Self> {
        // the library copies ss_params / ss_engine_opts by value: refuse a library whose layouts are not the ones declared above (speaksense.h SS_ABI_VERSION)
        let (v, sp, so) = unsafe { (ss_abi_version(), ss_sizeof_params(), ss_sizeof_engine_opts()) };
        if v != 6 || sp as usize != std::mem::size_of::<ss_params>() || so as usize != std::mem::size_of::<ss_engine_opts>() {
            return Err(anyhow!("libspeaksense_hip ABI {} (ss_params {} B, ss_engine_opts {} B) does not match this shim (ABI 6, {} B, {} B)", v, sp, so,
                               std::mem::size_of::<ss_params>(), std::mem::size_of::<ss_engine_opts>()));
        }
        let path = CString::new(model_path)?;
        let opts = ss_engine_opts { device: 0, dtype: 1 /* SS_DTYPE_F16 (ggml's arithmetic); 0 = bf16, 2 = fp8 (e4m3 encoder / cross-KV projections and cross cache, base and larger models) */, max_batch: 8, max_decoders: 5, batch_wait_us: 2000, n_lanes: 2, compat: 0 /* whisper.cpp v1.5.x behaviour (what whisper-rs-sys 0.9.0 vendors); SS_COMPAT_* selects older / OpenAI variants */, reserved: 0 };
        let mut e: *mut ss_engine = std::ptr::null_mut();
        let rc = unsafe { ss_engine_create(path.as_ptr(), &opts, &mut e) };
        if rc != 0 { return Err(anyhow!("failed to open whisper model: {}", last_error())); }
        Ok(Self { engine: Arc::new(EnginePtr(e)) })
    }

    pub fn create_state(&self) -> Result<Arc<Mutex<Box<HipSession>>>> {
        let s = unsafe { ss_session_create(self.engine.0) };
        if s.is_null() { return Err(anyhow!("Failed to create whisper state")); }
        Ok(Arc::new(Mutex::new(Box::new(WhisperState { raw: s, _ctx: PhantomData }))))
    }

    fn build_params(&self, ap: &AsrParams) -> ss_params {
        let mut p: ss_params = unsafe { std::mem::zeroed() };
        unsafe { ss_default_params(&mut p) };
        p.tdrz_enable = ap.speaker_diarization as i32;
        p.no_context = 0;
        if let Some(lang) = &ap.language {
            let b = lang.as_bytes();
            p.language = [0; 8];
            p.language[..b.len().min(7)].copy_from_slice(&b[..b.len().min(7)]);
        }
        if ap.stream_mode { p.single_segment = 0; p.no_context = 1; p.audio_ctx = 0; }
        p
    }

    fn is_promotional_text(text: &str) -> bool { PROMOTIONAL_TEXT.iter().any(|&p| text.contains(p)) }

    fn add_punctuation(text: &str) -> String {
        if text.ends_with(['。', '！', '？', '，']) { return text.to_string(); }
        let q = ["吗", "呢", "什么", "为何", "怎么"].iter().any(|k| text.contains(k));
        let e = ["啊", "哇", "太", "真", "好", "真是"].iter().any(|k| text.contains(k));
        let mut r = String::with_capacity(text.len() + 3);
        r.push_str(text);
        r.push(if q { '？' } else if e { '！' } else { ' ' });
        r
    }

    pub async fn transcribe_with_state(&self, state: Arc<Mutex<Box<HipSession>>>, audio: Vec<f32>, user_params: AsrParams) -> Result<TranscribeResult> {
        let p = self.build_params(&user_params);
        let stream_mode = user_params.stream_mode;
        // the GPU call never runs on a tokio worker: submit is non-blocking, wait happens on the blocking pool
        tokio::task::spawn_blocking(move || -> Result<TranscribeResult> {
            let guard = state.lock().map_err(|e| anyhow!("Failed to lock state: {}", e))?;
            let s = guard.raw;
            let mut t: *mut ss_ticket = std::ptr::null_mut();
            let rc = unsafe { ss_submit(s, audio.as_ptr(), audio.len() as i32, &p, &mut t) };
            if rc != 0 { return Err(anyhow!("submit failed: {}", last_error())); }
            let rc = unsafe { ss_wait(t) };
            if rc != 0 { return Err(anyhow!("transcription failed ({}): {}", rc, last_error())); }
            let n = unsafe { ss_result_n_segments(s) };
            let (mut segments, mut full_text, mut speaker) = (Vec::new(), String::new(), 0usize);
            for i in 0..n {
                let text = unsafe { CStr::from_ptr(ss_result_segment_text(s, i)) }.to_str()?.to_owned();  // strict UTF-8 as whisper.rs:85
                if Self::is_promotional_text(&text) { continue; }
                let (t0, t1) = unsafe { (ss_result_segment_t0(s, i), ss_result_segment_t1(s, i)) };
                if i > 0 && unsafe { ss_result_segment_speaker_turn_next(s, i - 1) } != 0 { speaker += 1; }
                let processed = Self::add_punctuation(&text);
                let seg = TranscribeSegment { text: processed.clone(), speaker_id: speaker, start: t0 as f64, end: t1 as f64 };
                if stream_mode { if i == n - 1 { segments.push(seg); full_text = processed; } }
                else { segments.push(seg); full_text.push_str(&processed); }
            }
            Ok(TranscribeResult { segments, full_text })
        }).await?
    }
}

// The trait the callers hold (`Arc<dyn AsrEngine>`, grpc/handlers/asr.rs:20-22,63-66): same three methods, same defaults (mod.rs:58-73).
#[async_trait::async_trait]
impl AsrEngine for HipAsr {
    fn create_state(&self) -> Result<Arc<Mutex<Box<WhisperState<'static>>>>> {
        HipAsr::create_state(self)
    }

    async fn transcribe_with_state(
        &self,
        state: Arc<Mutex<Box<WhisperState<'static>>>>,
        audio: Vec<f32>,
        params: AsrParams,
    ) -> Result<TranscribeResult> {
        HipAsr::transcribe_with_state(self, state, audio, params).await
    }

    async fn transcribe(&self, audio: Vec<f32>, params: AsrParams) -> Result<TranscribeResult> {
        let state = HipAsr::create_state(self)?;
        HipAsr::transcribe_with_state(self, state, audio, params).await
    }
}
