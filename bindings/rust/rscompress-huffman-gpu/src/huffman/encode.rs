//! `Encoder`: `std::io::Write` adaptor whose bit packing runs on the GPU
//! (reference: src/huffman/encode.rs).  `write` stages the bytes, `flush` (or a
//! full stage) encodes them: the reference's progressive inner writes are not
//! observable in the final stream.  The bytes handed to `inner` are the
//! reference's bytes; `readbytes`, `fillbits` and, after `flush`, `writeout`
//! are its values; `write`'s return value follows its put/cleanup schedule
//! through counters while `track_write_value` is set.
//!
//! Like the reference's, `flush` does not reset the encoder (encode.rs:118-136):
//! the bytes that were still in the reference's 64-bit buffer are emitted by the
//! flush and stay in the buffer, so a second `flush` emits them again and a
//! `write` after a `flush` continues in front of them.  Which bytes those are is
//! part of the schedule: once `track_write_value` was clear at a `write`, an
//! encoder that has been flushed refuses further use.  The gap array describes
//! the one continuous bit stream; it is dropped when symbols follow a flush that
//! ended inside a segment.
use crate::gpu::{io_check, with_ctx, DeviceModel};
use crate::model::Model;
use hfgpu_sys as sys;
use std::io::{Error, ErrorKind, Write};

/// Side metadata of a stream: bit length of every chunk of `chunk_syms` symbols
/// and the bit offset of every 256th chunk.  Lives beside the unchanged stream.
#[derive(Clone, Default)]
pub struct GapArray {
    pub chunk_syms: u32,
    pub chunk_bits: Vec<u16>,
    pub seg_off: Vec<u64>,
    pub symbols: u64,
}

/// Bytes staged before the device encodes them (whole segments, so the gap
/// arrays of consecutive batches concatenate).
const STAGE_LIMIT: usize = 256 << 20;

pub struct Encoder<'a, W: Write, M: Model> {
    pub inner: W,
    pub model: &'a M,
    pub fillbits: Option<u8>,
    pub readbytes: usize,
    pub writeout: usize,
    /// `write` returns the reference's value (one pass over the symbols on the
    /// host); clear it when the value is not used.
    pub track_write_value: bool,
    pub gap: GapArray,
    device: Option<DeviceModel>,
    pending: Vec<u8>,
    total_bits: u64,
    carry: u8,
    // whole bytes of the reference's buffer at the last flush: out again first
    resend: Vec<u8>,
    flushed: bool,
    gap_kept: bool,
    schedule_valid: bool, // every symbol so far went through advance_schedule
    // the reference's {buffer, remaining_bits} (encode.rs:43-47), counters only
    buffer: u64,
    remaining_bits: u64,
    schedule_writeout: usize,
}

impl<'a, W: Write, M: Model> Encoder<'a, W, M> {
    pub fn new(writer: W, model: &'a M) -> Self {
        Encoder {
            inner: writer, model, fillbits: None, readbytes: 0, writeout: 0,
            track_write_value: true,
            gap: GapArray { chunk_syms: 64, ..Default::default() },
            device: None, pending: Vec::new(), total_bits: 0, carry: 0,
            resend: Vec::new(), flushed: false, gap_kept: true, schedule_valid: true,
            buffer: 0, remaining_bits: 64, schedule_writeout: 0,
        }
    }
    pub fn magic(&self) -> Vec<u8> {
        b"pzhf".to_vec()
    }
    pub fn plain_write(&mut self, bytes: &[u8]) -> std::io::Result<()> {
        self.inner.write_all(bytes)
    }

    fn device(&mut self) -> std::io::Result<&DeviceModel> {
        if self.device.is_none() {
            self.device = Some(DeviceModel::from_model(self.model)?);
        }
        Ok(self.device.as_ref().unwrap())
    }

    /// Bytes `put` (one) and `cleanup` (five) would have emitted for `buf`, and
    /// the value `write` returns (encode.rs:57-83, :115).
    fn advance_schedule(&mut self, buf: &[u8]) -> std::io::Result<usize> {
        let (codes, lens) = {
            let d = self.device()?;
            (d.codes, d.lens)
        };
        let mut emitted = 0usize;
        for &sym in buf {
            let (code, len) = (codes[sym as usize], lens[sym as usize]);
            if len > 64 {
                return Err(Error::new(ErrorKind::InvalidData, "Codelen > 64"));
            }
            while len > self.remaining_bits {
                emitted += 1;
                self.buffer <<= 8;
                self.remaining_bits += 8;
            }
            self.remaining_bits -= len;
            self.buffer = self.buffer.wrapping_add(code << (self.remaining_bits & 63));
            if self.buffer & 0x0000_0000_00FF_0000 != 0 {
                emitted += 5;
                self.buffer <<= 40;
                self.remaining_bits += 40;
            }
        }
        self.schedule_writeout += emitted;
        Ok(emitted + 8 - (self.buffer.trailing_zeros() / 8) as usize)
    }

    fn refuse_after_untracked_flush(&self) -> std::io::Result<()> {
        if self.flushed && !self.schedule_valid {
            return Err(Error::new(ErrorKind::Other,
                "Encoder used again after flush(): which bytes the reference emits again is only known with track_write_value set"));
        }
        Ok(())
    }

    /// Encodes what is staged; whole bytes go to `inner`, a trailing partial
    /// byte is kept and merged into the next batch as lead bits -- after a
    /// flush, which emits it, too.
    fn drain(&mut self, last: bool) -> std::io::Result<usize> {
        let seg_syms = sys::HF_SEG_CHUNKS as usize * self.gap.chunk_syms as usize;
        let n = if last { self.pending.len() } else { self.pending.len() / seg_syms * seg_syms };
        if n == 0 && !last {
            return Ok(0);
        }
        let mut handed_on = 0usize;
        if !self.resend.is_empty() {
            let again = std::mem::take(&mut self.resend);
            self.inner.write_all(&again)?;
            handed_on += again.len();
        }
        if n > 0 && self.gap.symbols as usize % seg_syms != 0 {
            self.gap = GapArray { chunk_syms: self.gap.chunk_syms, ..Default::default() };
            self.gap_kept = false;
        }
        let lead = (self.total_bits % 8) as u32;
        let handle = self.device()?.handle;
        let cap = unsafe { sys::hf_encode_bound(handle, n as u64, lead) } as usize + 8;
        let mut out: Vec<u8> = Vec::with_capacity(cap);
        let (mut nch, mut nsg) = (0u64, 0u64);
        if self.gap_kept {
            unsafe { sys::hf_gap_layout(n as u64, self.gap.chunk_syms, &mut nch, &mut nsg) };
        }
        let mut cb = vec![0u16; nch as usize];
        let mut so = vec![0u64; nsg as usize];
        let mut bits = 0u64;
        if n > 0 {
            let with_gap = self.gap_kept;
            with_ctx(|ctx| {
                let rc = unsafe {
                    sys::hf_encode_host(ctx, handle, self.pending.as_ptr(), n as u64, lead,
                                        out.as_mut_ptr(), cap as u64, self.gap.chunk_syms,
                                        if with_gap { cb.as_mut_ptr() } else { std::ptr::null_mut() },
                                        if with_gap { so.as_mut_ptr() } else { std::ptr::null_mut() },
                                        &mut bits)
                };
                io_check(ctx, rc, "Encoder::flush")
            })?;
        } else {
            // a flush with nothing staged: only the partial byte, if there is one
            unsafe { *out.as_mut_ptr() = 0 };
        }
        let mut nbytes = ((lead as u64 + bits + 7) / 8) as usize;
        unsafe { out.set_len(nbytes) };
        if self.gap_kept {
            let base_bit = self.total_bits - lead as u64;
            self.gap.chunk_bits.extend_from_slice(&cb);
            self.gap.seg_off.extend(so.iter().map(|v| v + base_bit));
            self.gap.symbols += n as u64;
        }
        if lead != 0 && nbytes > 0 {
            out[0] |= self.carry;
        }
        self.total_bits += bits;
        let partial = self.total_bits % 8 != 0;
        if partial && nbytes > 0 {
            self.carry = out[nbytes - 1];
        }
        if !last && partial && nbytes > 0 {
            nbytes -= 1;
        }
        self.inner.write_all(&out[..nbytes])?;
        handed_on += nbytes;
        if last {
            self.fillbits = Some(((8 - self.total_bits % 8) % 8) as u8);
            self.flushed = true;
            if self.schedule_valid {
                // encode.rs:119-133: what the flush has just emitted from the reference's
                // buffer stays there; its whole bytes go out again before anything new
                let held = self.buffer.to_be_bytes();
                let length = 8 - (self.remaining_bits / 8) as usize;
                self.resend = held[..length - partial as usize].to_vec();
                self.schedule_writeout += length;
                self.writeout = self.schedule_writeout;
            } else {
                self.writeout = ((self.total_bits + 7) / 8) as usize;
            }
        } else if !self.track_write_value || !self.schedule_valid {
            self.writeout = ((self.total_bits + 7) / 8) as usize;
        }
        self.pending.drain(..n);
        Ok(handed_on)
    }
}

impl<'a, W: Write, M: Model> Write for Encoder<'a, W, M> {
    fn write(&mut self, buf: &[u8]) -> std::io::Result<usize> {
        self.device()?;
        self.refuse_after_untracked_flush()?;
        let mut ret = 0usize;
        let replay = self.track_write_value && self.schedule_valid;
        if replay {
            ret = self.advance_schedule(buf)?;
            self.writeout = self.schedule_writeout;
        } else if !buf.is_empty() {
            self.schedule_valid = false;
        }
        self.readbytes += buf.len();
        if self.pending.len() + buf.len() > (1 << 20) && self.pending.capacity() < STAGE_LIMIT {
            self.pending.reserve(STAGE_LIMIT + buf.len() - self.pending.len());
        }
        self.pending.extend_from_slice(buf);
        let mut emitted = 0usize;
        if self.pending.len() >= STAGE_LIMIT {
            emitted = self.drain(false)?;
        }
        Ok(if replay { ret } else { emitted })
    }
    fn flush(&mut self) -> std::io::Result<()> {
        self.device()?;
        self.refuse_after_untracked_flush()?;
        self.drain(true)?;
        self.inner.flush()
    }
}

/// Bit length of `val`; 0 and 1 both take one bit (encode.rs:140-149).
pub fn calculate_length(val: usize) -> usize {
    unsafe { sys::hf_calculate_length(val as u64) as usize }
}
