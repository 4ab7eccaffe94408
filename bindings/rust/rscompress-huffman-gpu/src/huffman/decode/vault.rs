//! The reference's experimental third decoder (decode/vault.rs).  Only its
//! results are reproduced -- the same bytes as `Decoder` -- together with its
//! precondition `sentinel <= 8` (vault.rs:52-56).
use super::Decoder as Streaming;
use crate::huffman::header::Header;
use crate::model::Model;
use std::io::Read;

pub struct Decoder<R: Read> {
    inner: Streaming<R>,
}

impl<R: Read> Decoder<R> {
    pub fn new<M: Model>(reader: R, model: &M, output: u64) -> Self {
        assert!(model.sentinel() <= 8);
        let header = Header {
            magic: b"pzhf".to_vec(),
            btree: model.to_btreemap(),
            sentinel: model.sentinel(),
            readbytes: output as usize,
        };
        Decoder { inner: Streaming::from_header(header, reader) }
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> std::io::Result<usize> {
        self.inner.read(buf)
    }
}
