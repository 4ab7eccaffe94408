//! Codebook construction (reference: src/huffman/mod.rs).  A problem of at most
//! 256 elements: it stays on the host.  The functions call the library's host
//! routines so that the Rust, C++ and Python hosts share one implementation
//! and one tie rule (equal counts keep ascending symbol order).
use crate::gpu::{check, with_ctx};
use crate::model::Model;
use hfgpu_sys as sys;

pub mod decode;
pub mod encode;
pub mod header;

pub struct Huffman {
    codewords: [usize; 256],
    length: [usize; 256],
}

impl Model for Huffman {
    fn encode(&self, sym: u8) -> (usize, usize) {
        (self.codewords[sym as usize], self.length[sym as usize])
    }
    fn sentinel(&self) -> usize {
        let lens: Vec<u64> = self.length.iter().map(|l| *l as u64).collect();
        unsafe { sys::hf_model_sentinel(lens.as_ptr()) as usize }
    }
    fn to_btreemap(&self) -> BTreeMap<usize, (u8, u8)> {
        let codes: Vec<u64> = self.codewords.iter().map(|c| *c as u64).collect();
        let lens: Vec<u64> = self.length.iter().map(|l| *l as u64).collect();
        let (mut keys, mut syms, mut klens, mut m) = ([0u64; 256], [0u8; 256], [0u8; 256], 0u64);
        let rc = unsafe {
            sys::hf_model_to_table(codes.as_ptr(), lens.as_ptr(), keys.as_mut_ptr(),
                                   syms.as_mut_ptr(), klens.as_mut_ptr(), &mut m)
        };
        assert_eq!(rc, sys::HF_OK, "to_btreemap");
        (0..m as usize).map(|i| (keys[i] as usize, (syms[i], klens[i]))).collect()
    }
}

use crate::stats::generate_histogram;
use std::collections::BTreeMap;
use std::io::{Read, Seek, SeekFrom};

impl Huffman {
    pub fn new(codewords: [usize; 256], length: [usize; 256]) -> Self {
        Huffman { codewords, length }
    }
    pub fn from_histogram(histogram: &[usize; 256]) -> Self {
        let (codewords, length) = generate_extended_codewords(histogram);
        Huffman::new(codewords, length)
    }
    pub fn from_reader<RS: Seek + Read>(reader: &mut RS) -> Self {
        let histogram = generate_histogram(reader);
        reader.seek(SeekFrom::Start(0)).expect("Can not move to start of file");
        Huffman::from_histogram(&histogram)
    }
    pub fn from_slice(data: &[u8]) -> Self {
        let mut hist = [0u64; 256];
        with_ctx(|ctx| {
            let rc = unsafe { sys::hf_histogram_host(ctx, data.as_ptr(), data.len() as u64, hist.as_mut_ptr()) };
            check(ctx, rc, "from_slice");
        });
        let mut histogram = [0usize; 256];
        for (h, v) in histogram.iter_mut().zip(hist.iter()) {
            *h = *v as usize;
        }
        Huffman::from_histogram(&histogram)
    }
}

fn to_u64(v: &[usize]) -> Vec<u64> {
    v.iter().map(|x| *x as u64).collect()
}

/// Weights in non-increasing order are replaced by code lengths (mod.rs:71-125).
pub fn calculate_codeword_length_inplace(histogram: &mut [usize]) {
    let mut w = to_u64(histogram);
    let rc = unsafe { sys::hf_codeword_lengths_inplace(w.as_mut_ptr(), w.len() as u64) };
    // the reference underflows an index on an empty slice (mod.rs:72-73)
    assert_eq!(rc, sys::HF_OK, "attempt to subtract with overflow");
    for (h, v) in histogram.iter_mut().zip(w.iter()) {
        *h = *v as usize;
    }
}

/// (index, count) of the non-zero counts, descending by count (mod.rs:127-141).
pub fn sort_by_value(store: &[usize]) -> Vec<(usize, usize)> {
    assert!(store.len() <= 256);
    let s = to_u64(store);
    let (mut idx, mut cnt, mut m) = ([0u64; 256], [0u64; 256], 0u64);
    let rc = unsafe { sys::hf_sort_by_value(s.as_ptr(), s.len() as u64, idx.as_mut_ptr(), cnt.as_mut_ptr(), &mut m) };
    assert_eq!(rc, sys::HF_OK);
    (0..m as usize).map(|i| (idx[i] as usize, cnt[i] as usize)).collect()
}

pub fn extract_values(store: &[(usize, usize)]) -> Vec<usize> {
    store.iter().map(|p| p.1).collect()
}

/// Canonical codes of non-decreasing lengths: (codes, left-aligned codes) (mod.rs:155-171).
pub fn calculate_codewords_based_on_length(lengths: &[usize]) -> (Vec<usize>, Vec<usize>) {
    let l = to_u64(lengths);
    let mut small = vec![0u64; l.len()];
    let mut big = vec![0u64; l.len()];
    let rc = unsafe { sys::hf_codewords_from_lengths(l.as_ptr(), l.len() as u64, small.as_mut_ptr(), big.as_mut_ptr()) };
    assert_eq!(rc, sys::HF_OK);
    (small.iter().map(|x| *x as usize).collect(), big.iter().map(|x| *x as usize).collect())
}

/// Histogram (any length up to 256) -> codes and lengths by symbol (mod.rs:181-206).
pub fn generate_extended_codewords(histogram: &[usize]) -> ([usize; 256], [usize; 256]) {
    let h = to_u64(histogram);
    let (mut codes, mut lens) = ([0u64; 256], [0u64; 256]);
    let rc = unsafe { sys::hf_build_codebook(h.as_ptr(), h.len() as u64, codes.as_mut_ptr(), lens.as_mut_ptr()) };
    assert_eq!(rc, sys::HF_OK, "generate_extended_codewords: empty or oversized histogram");
    let (mut c, mut l) = ([0usize; 256], [0usize; 256]);
    for i in 0..256 {
        c[i] = codes[i] as usize;
        l[i] = lens[i] as usize;
    }
    (c, l)
}
