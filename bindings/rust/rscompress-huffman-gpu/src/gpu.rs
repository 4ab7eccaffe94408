//! Process-wide context of the GPU library and the status -> Rust error mapping.
//! Not part of the reference's API: everything here is crate-private plumbing.
use hfgpu_sys as sys;
use std::ffi::CStr;
use std::io::{Error, ErrorKind};
use std::sync::{Mutex, OnceLock};

pub(crate) struct Ctx(pub *mut sys::hf_ctx);
// the handle is only ever used under the mutex below
unsafe impl Send for Ctx {}

static CTX: OnceLock<Mutex<Ctx>> = OnceLock::new();

/// One context on device 0 (HFGPU_DEVICE overrides), created on first use.
/// Calls on a context must not overlap: the mutex serialises them.
pub(crate) fn with_ctx<T>(f: impl FnOnce(*mut sys::hf_ctx) -> T) -> T {
    let cell = CTX.get_or_init(|| {
        let device = std::env::var("HFGPU_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let mut h: *mut sys::hf_ctx = std::ptr::null_mut();
        let rc = unsafe { sys::hf_ctx_create(device, &mut h) };
        if rc != sys::HF_OK {
            panic!("hf_ctx_create failed: a GPU is required, there is no CPU path");
        }
        Mutex::new(Ctx(h))
    });
    let guard = cell.lock().expect("hfgpu context poisoned");
    f(guard.0)
}

pub(crate) fn last_error(ctx: *mut sys::hf_ctx) -> String {
    unsafe { CStr::from_ptr(sys::hf_last_error(ctx)).to_string_lossy().into_owned() }
}

/// Non-zero status -> panic, as the reference panics (`expect`/`unwrap`) on
/// the same conditions; used where the reference's signature has no Result.
pub(crate) fn check(ctx: *mut sys::hf_ctx, status: i32, what: &str) {
    if status != sys::HF_OK {
        let name = unsafe { CStr::from_ptr(sys::hf_strerror(status)).to_string_lossy() };
        panic!("{}: {} ({})", what, name, last_error(ctx));
    }
}

/// Non-zero status -> io::Error; a code length above 64 keeps the reference's
/// kind and text (src/huffman/encode.rs:102-104).
pub(crate) fn io_check(ctx: *mut sys::hf_ctx, status: i32, what: &str) -> std::io::Result<()> {
    match status {
        sys::HF_OK => Ok(()),
        sys::HF_ERR_CODELEN => Err(Error::new(ErrorKind::InvalidData, "Codelen > 64")),
        _ => Err(Error::new(ErrorKind::Other, format!("{}: {}", what, last_error(ctx)))),
    }
}

/// Device-resident tables of a `Model` (hf_model), destroyed with the value.
pub(crate) struct DeviceModel {
    pub handle: *mut sys::hf_model,
    pub codes: [u64; 256],
    pub lens: [u64; 256],
}

impl DeviceModel {
    pub fn from_model<M: crate::model::Model>(model: &M) -> std::io::Result<Self> {
        let mut codes = [0u64; 256];
        let mut lens = [0u64; 256];
        for s in 0..256usize {
            let (c, l) = model.encode(s as u8);
            codes[s] = c as u64;
            lens[s] = l as u64;
        }
        let mut handle = std::ptr::null_mut();
        with_ctx(|ctx| {
            let rc = unsafe { sys::hf_model_create(ctx, codes.as_ptr(), lens.as_ptr(), &mut handle) };
            io_check(ctx, rc, "Huffman model")
        })?;
        Ok(DeviceModel { handle, codes, lens })
    }
    pub fn from_table(btree: &std::collections::BTreeMap<usize, (u8, u8)>, sentinel: usize) -> Self {
        let keys: Vec<u64> = btree.keys().map(|k| *k as u64).collect();
        let syms: Vec<u8> = btree.values().map(|v| v.0).collect();
        let klens: Vec<u8> = btree.values().map(|v| v.1).collect();
        let mut handle = std::ptr::null_mut();
        with_ctx(|ctx| {
            let rc = unsafe {
                sys::hf_model_from_table(ctx, keys.as_ptr(), syms.as_ptr(), klens.as_ptr(),
                                         keys.len() as u64, sentinel as u64, &mut handle)
            };
            check(ctx, rc, "decode table");
        });
        DeviceModel { handle, codes: [0; 256], lens: [0; 256] }
    }
}

impl Drop for DeviceModel {
    fn drop(&mut self) {
        unsafe { sys::hf_model_destroy(self.handle) };
    }
}
