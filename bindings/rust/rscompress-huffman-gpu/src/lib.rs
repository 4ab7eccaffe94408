//! The public API of the rscompress-huffman crate with its hot path --
//! histogram, bit packing, decoding -- on an AMD MI355X through `libhfgpu.so`
//! (declared in the `hfgpu-sys` crate).  Static, two-pass, canonical Huffman
//! coding of bytes; the files it writes are the reference's files.
//!
//! There is no CPU codec in here: without the library and a GPU every entry
//! point panics or returns an error.
use log::info;
use std::fs::File;
use std::io::prelude::*;
use std::io::{BufReader, BufWriter, SeekFrom};

pub const BUF: usize = 4096;
pub mod dist;
mod gpu;
pub mod huffman;
pub mod model;
pub mod stats;

use crate::huffman::decode::Decoder;
use crate::huffman::encode::{Encoder, GapArray};
use crate::huffman::header::Header;
use crate::huffman::Huffman;

fn gap_sidecar_path(path: &str) -> String {
    format!("{}.hfgap", path)
}

/// "hfgap1\0\0", u32 chunk_syms, u64 symbols, u64 n_chunks, u64 n_segs,
/// chunk_bits, seg_off (little endian) -- beside the unchanged stream.
fn write_gap_sidecar(path: &str, g: &GapArray) -> std::io::Result<()> {
    let mut f = BufWriter::new(File::create(path)?);
    f.write_all(b"hfgap1\0\0")?;
    f.write_all(&g.chunk_syms.to_le_bytes())?;
    f.write_all(&g.symbols.to_le_bytes())?;
    f.write_all(&(g.chunk_bits.len() as u64).to_le_bytes())?;
    f.write_all(&(g.seg_off.len() as u64).to_le_bytes())?;
    for b in &g.chunk_bits {
        f.write_all(&b.to_le_bytes())?;
    }
    for o in &g.seg_off {
        f.write_all(&o.to_le_bytes())?;
    }
    f.flush()
}

fn read_gap_sidecar(path: &str) -> Option<GapArray> {
    let bytes = std::fs::read(path).ok()?;
    if bytes.len() < 36 || &bytes[..6] != b"hfgap1" {
        return None;
    }
    let u64_at = |o: usize| u64::from_le_bytes([bytes[o], bytes[o + 1], bytes[o + 2], bytes[o + 3],
                                               bytes[o + 4], bytes[o + 5], bytes[o + 6], bytes[o + 7]]);
    let chunk_syms = u32::from_le_bytes([bytes[8], bytes[9], bytes[10], bytes[11]]);
    let (symbols, nc, ns) = (u64_at(12), u64_at(20), u64_at(28));
    // the counts must account for the file exactly
    if nc > bytes.len() as u64 / 2 || ns > bytes.len() as u64 / 8 || 36 + nc * 2 + ns * 8 != bytes.len() as u64 {
        return None;
    }
    let cb0 = 36usize;
    let so0 = cb0 + 2 * nc as usize;
    Some(GapArray {
        chunk_syms,
        symbols,
        chunk_bits: (0..nc as usize).map(|i| u16::from_le_bytes([bytes[cb0 + 2 * i], bytes[cb0 + 2 * i + 1]])).collect(),
        seg_off: (0..ns as usize).map(|i| u64_at(so0 + 8 * i)).collect(),
    })
}

/// `unimplemented!()` in the reference (src/lib.rs:28-30); completed through
/// `Header::from_binary` + `Decoder::from_header`.  The side file is used only
/// if it fits this very stream; without it the device finds the code
/// boundaries itself.
pub fn stream_decompress_with_header_information(source: &str, destination: &str) {
    let file = File::open(source).expect("Failed to open source file");
    let file_size = file.metadata().expect("metadata").len();
    let mut reader = BufReader::with_capacity(1 << 20, file);
    let mut fixed = [0u8; 12];
    reader.read_exact(&mut fixed).expect("not a pzhf file");
    assert_eq!(&fixed[..4], b"pzhf", "not a pzhf file");
    let mut be = [0u8; 8];
    be.copy_from_slice(&fixed[4..]);
    let hlen = bytes_to_u64(be);
    assert!(hlen <= file_size - 12, "header length exceeds the file");
    let mut hb = vec![0u8; hlen as usize];
    reader.read_exact(&mut hb).expect("truncated header");
    let header = Header::from_binary(&hb);
    let gap = read_gap_sidecar(&gap_sidecar_path(source));
    let mut decoder = Decoder::from_header_with_gap(header, reader, gap, Some(file_size - 12 - hlen));
    let mut out = BufWriter::with_capacity(1 << 20, File::create(destination).expect("Failed to create destination file"));
    std::io::copy(&mut decoder, &mut out).expect("Failed to write");
    out.flush().expect("Failed to flush");
}

/// "pzhf" || u64 big-endian header length || bincode(Header) || bitstream
/// (src/lib.rs:31-80), plus the gap side file `<destination>.hfgap`.
pub fn stream_compress_with_header_information(source: &str, destination: &str) {
    let file = File::open(source).expect("Failed to open source file");
    let filesize = file.metadata().expect("metadata").len();
    let mut reader = BufReader::with_capacity(1 << 20, file);
    info!("Generating model");
    let model = Huffman::from_reader(&mut reader);
    reader.seek(SeekFrom::Start(0)).expect("Can not move to start of file");
    let out = BufWriter::with_capacity(1 << 20, File::create(destination).expect("Failed to create destination file"));
    let mut writer = Encoder::new(out, &model);
    writer.track_write_value = false; // write()'s return value is not used here
    let mut header = Header::from(&writer);
    header.update_readbytes(filesize);
    let header_bytes = header.to_binary();
    let magic = writer.magic();
    writer.plain_write(&magic).expect("Failed to write magic");
    writer.plain_write(&u64_to_bytes(header_bytes.len() as u64)).expect("Failed to write header length");
    writer.plain_write(&header_bytes).expect("Failed to write header");
    let mut buffer = vec![0u8; BUF];
    loop {
        let n = reader.read(&mut buffer).expect("Failed to read source");
        if n == 0 {
            break;
        }
        writer.write_all(&buffer[..n]).expect("Failed to encode");
    }
    writer.flush().expect("Failed to flush");
    write_gap_sidecar(&gap_sidecar_path(destination), &writer.gap).expect("Failed to write the side file");
}

/// Big-endian bytes of `num` (src/lib.rs:82-93).
pub fn u64_to_bytes(num: u64) -> [u8; 8] {
    num.to_be_bytes()
}
pub fn bytes_to_u64(bytes: [u8; 8]) -> u64 {
    u64::from_be_bytes(bytes)
}
