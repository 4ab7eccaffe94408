//! Decoders (reference: src/huffman/decode/mod.rs).  Lookup is the reference's
//! predecessor search over the model's keys, done by the device tables, so a
//! model that is not prefix free decodes as it does there.
use super::encode::{Encoder, GapArray};
use super::header::Header;
use crate::gpu::{check, with_ctx, DeviceModel};
use crate::model::Model;
use hfgpu_sys as sys;
use std::collections::BTreeMap;
use std::io::{Read, Write};

pub mod vault;

/// Symbols per piece with a gap array / stream bytes per window without one.
const PIECE_SYMS: u64 = 64 << 20;
const WINDOW_BYTES: usize = 32 << 20;

/// Streaming decoder: `read` hands out decoded bytes piece by piece; the host
/// holds a window of the stream and one piece of output.
pub struct Decoder<R: Read> {
    inner: R,
    goalsbyte: u64,
    device: DeviceModel,
    gap: Option<GapArray>,
    comp: Vec<u8>,
    comp_base: u64,
    eof: bool,
    next_sym: u64,
    start_bit: u32,
    out: Vec<u8>,
    pos: usize,
}

/// Whether side metadata fits a stream of `readbytes` symbols: sizes, first
/// offset, every segment offset = predecessor + chunk lengths in between and,
/// when known, the last bit inside the stream's last byte.
pub fn gap_fits(g: &GapArray, readbytes: u64, stream_bytes: Option<u64>) -> bool {
    let (mut nc, mut ns) = (0u64, 0u64);
    if g.symbols != readbytes || unsafe { sys::hf_gap_layout(readbytes, g.chunk_syms, &mut nc, &mut ns) } != sys::HF_OK {
        return false;
    }
    if g.chunk_bits.len() as u64 != nc || g.seg_off.len() as u64 != ns {
        return false;
    }
    let mut run = 0u64;
    for (sg, off) in g.seg_off.iter().enumerate() {
        if *off != run {
            return false;
        }
        let c0 = sg * sys::HF_SEG_CHUNKS as usize;
        let c1 = (c0 + sys::HF_SEG_CHUNKS as usize).min(nc as usize);
        run += g.chunk_bits[c0..c1].iter().map(|b| *b as u64).sum::<u64>();
    }
    stream_bytes.map_or(true, |b| (run + 7) / 8 == b)
}

impl<R: Read> Decoder<R> {
    pub fn new<W: Write, M: Model>(reader: R, encoder: &Encoder<W, M>) -> Self {
        let device = DeviceModel::from_model(encoder.model).expect("decode table");
        let goals = encoder.readbytes as u64;
        let gap = Some(encoder.gap.clone()).filter(|g| goals > 0 && gap_fits(g, goals, None));
        Decoder::with(reader, goals, device, gap)
    }
    pub fn from_header(header: Header, reader: R) -> Self {
        Decoder::from_header_with_gap(header, reader, None, None)
    }
    /// `from_header` with the side file's content and, if known, the stream's length.
    pub fn from_header_with_gap(header: Header, reader: R, gap: Option<GapArray>,
                                stream_bytes: Option<u64>) -> Self {
        let goals = header.readbytes as u64;
        let device = DeviceModel::from_table(&header.btree, header.sentinel);
        let gap = gap.filter(|g| goals > 0 && gap_fits(g, goals, stream_bytes));
        Decoder::with(reader, goals, device, gap)
    }
    fn with(inner: R, goalsbyte: u64, device: DeviceModel, gap: Option<GapArray>) -> Self {
        Decoder { inner, goalsbyte, device, gap, comp: Vec::new(), comp_base: 0, eof: false,
                  next_sym: 0, start_bit: 0, out: Vec::new(), pos: 0 }
    }

    fn fill_to(&mut self, upto: u64) -> std::io::Result<()> {
        while !self.eof && self.comp_base + (self.comp.len() as u64) < upto {
            let have = self.comp.len();
            let want = ((upto - self.comp_base) as usize - have).min(1 << 22);
            self.comp.resize(have + want, 0);
            let n = self.inner.read(&mut self.comp[have..])?;
            self.comp.truncate(have + n);
            self.eof = n == 0;
        }
        Ok(())
    }
    fn drop_before(&mut self, byte: u64) {
        if byte > self.comp_base {
            let k = ((byte - self.comp_base) as usize).min(self.comp.len());
            self.comp.drain(..k);
            self.comp_base += k as u64;
        }
    }
    fn next_piece(&mut self) -> std::io::Result<bool> {
        self.out.clear();
        self.pos = 0;
        if self.next_sym >= self.goalsbyte {
            return Ok(false);
        }
        if self.gap.is_some() { self.piece_with_gap() } else { self.piece_from_window() }
    }
    fn piece_with_gap(&mut self) -> std::io::Result<bool> {
        // (taken for the duration of the call so that `self` can be borrowed mutably
        // below; put back on every way out)
        let g = self.gap.take().unwrap();
        let r = self.piece_with_gap_of(&g);
        self.gap = Some(g);
        r
    }
    fn piece_with_gap_of(&mut self, g: &GapArray) -> std::io::Result<bool> {
        let seg_syms = sys::HF_SEG_CHUNKS * g.chunk_syms as u64;
        let (s0, n_segs) = ((self.next_sym / seg_syms) as usize, g.seg_off.len());
        let s1 = n_segs.min(s0 + (PIECE_SYMS / seg_syms).max(1) as usize);
        let n_piece = self.goalsbyte.min(s1 as u64 * seg_syms) - self.next_sym;
        let total: u64 = g.seg_off[n_segs - 1]
            + g.chunk_bits[(n_segs - 1) * sys::HF_SEG_CHUNKS as usize..].iter().map(|b| *b as u64).sum::<u64>();
        let end_bit = if s1 < n_segs { g.seg_off[s1] } else { total };
        let first_byte = (g.seg_off[s0] / 8) & !15u64;
        self.drop_before(first_byte);
        self.fill_to((end_bit + 7) / 8 + 16)?;
        if self.comp_base != first_byte || self.comp_base + (self.comp.len() as u64) < (end_bit + 7) / 8 {
            return Err(std::io::Error::new(std::io::ErrorKind::UnexpectedEof,
                                           "the stream ends before the gap array does"));
        }
        self.out.resize(n_piece as usize, 0);
        with_ctx(|ctx| {
            let rc = unsafe {
                sys::hf_decode_range_host(ctx, self.device.handle, self.comp.as_ptr(),
                                          self.comp.len() as u64, first_byte * 8, n_piece, g.chunk_syms,
                                          g.chunk_bits[s0 * sys::HF_SEG_CHUNKS as usize..].as_ptr(),
                                          g.seg_off[s0..].as_ptr(), self.out.as_mut_ptr())
            };
            check(ctx, rc, "Decoder::read");
        });
        self.next_sym += n_piece;
        Ok(true)
    }
    fn piece_from_window(&mut self) -> std::io::Result<bool> {
        let mut window = WINDOW_BYTES;
        loop {
            self.fill_to(self.comp_base + window as u64)?;
            let left = self.goalsbyte - self.next_sym;
            let n_max = left.min(self.comp.len() as u64 * 8 + 8).min(256 << 20);
            let all = self.eof && n_max == left;
            self.out.resize(n_max as usize, 0);
            let (mut n, mut end_bit) = (0u64, 0u64);
            with_ctx(|ctx| {
                let rc = unsafe {
                    sys::hf_decode_window_host(ctx, self.device.handle, self.comp.as_ptr(),
                                               self.comp.len() as u64, self.start_bit, n_max,
                                               all as i32, self.out.as_mut_ptr(), &mut n, &mut end_bit)
                };
                check(ctx, rc, "Decoder::read");
            });
            if n == 0 && !all {
                assert!(!self.eof, "window without a whole segment");
                window *= 2;
                continue;
            }
            self.out.truncate(n as usize);
            self.next_sym += n;
            self.start_bit = (end_bit % 8) as u32;
            self.drop_before(self.comp_base + end_bit / 8);
            return Ok(n != 0);
        }
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> std::io::Result<usize> {
        let mut done = 0;
        while done < buf.len() {
            if self.pos == self.out.len() && !self.next_piece()? {
                break;
            }
            let k = (buf.len() - done).min(self.out.len() - self.pos);
            buf[done..done + k].copy_from_slice(&self.out[self.pos..self.pos + k]);
            self.pos += k;
            done += k;
        }
        Ok(done)
    }
}

/// Values of the map in key order plus the sorted keys.  The reference returns
/// the keys as a rank dictionary over a bit vector (`succinct::RsDict`,
/// decode/mod.rs:140-155); `rank1(v + 1) - 1` there is the index of the largest
/// key <= v here (`keys.partition_point(|k| *k <= v) - 1`).
pub fn prepare_lookup(bt: &BTreeMap<usize, (u8, u8)>) -> (Vec<(u8, u8)>, Vec<usize>) {
    (bt.values().cloned().collect(), bt.keys().cloned().collect())
}

/// Entry of the largest key <= `key` (decode/mod.rs:157-165); panics below the smallest key.
pub fn search_key_or_next_small_key(tree: &BTreeMap<usize, (u8, u8)>, key: usize) -> (u8, u8) {
    match tree.range(..=key).next_back() {
        Some((_, v)) => *v,
        None => panic!("Panic!!!!"),
    }
}

/// Slice decoder (decode/mod.rs:167-231): `goalsbyte` symbols of `data`.
pub fn read(data: &[u8], model: &impl Model, goalsbyte: usize) -> Vec<u8> {
    read_with_gap(data, model, goalsbyte, None)
}

/// `read` driven by the encoder's gap array (used only if it fits the stream).
pub fn read_with_gap(data: &[u8], model: &impl Model, goalsbyte: usize, gap: Option<&GapArray>) -> Vec<u8> {
    let device = DeviceModel::from_model(model).expect("decode table");
    let mut out = vec![0u8; goalsbyte];
    let gap = gap.filter(|g| goalsbyte > 0 && gap_fits(g, goalsbyte as u64, Some(data.len() as u64)));
    with_ctx(|ctx| {
        let rc = unsafe {
            sys::hf_decode_host(ctx, device.handle, data.as_ptr(), data.len() as u64, goalsbyte as u64,
                                gap.map_or(64, |g| g.chunk_syms),
                                gap.map_or(std::ptr::null(), |g| g.chunk_bits.as_ptr()),
                                gap.map_or(std::ptr::null(), |g| g.seg_off.as_ptr()), out.as_mut_ptr())
        };
        check(ctx, rc, "decode::read");
    });
    out
}
