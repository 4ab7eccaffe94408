//! Byte statistics (reference: src/stats.rs).  The counting runs on the GPU.
use crate::gpu::{check, with_ctx};
use hfgpu_sys as sys;
use std::io::Read;

/// Bytes handed to the device per call: the host holds one piece of the reader
/// whatever its length.
const PIECE: usize = 64 << 20;

/// 256-bin histogram of everything `reader` delivers (reference: stats.rs:13-28;
/// a read error panics there as well, stats.rs:24).
pub fn generate_histogram(reader: &mut impl Read) -> [usize; 256] {
    let mut hist = [0usize; 256];
    let mut piece = vec![0u8; PIECE.min(1 << 20)];
    loop {
        let mut have = 0usize;
        loop {
            if have == piece.len() {
                if piece.len() >= PIECE {
                    break;
                }
                piece.resize((piece.len() * 2).min(PIECE), 0);
            }
            let n = reader.read(&mut piece[have..]).expect("Failed to read source");
            if n == 0 {
                break;
            }
            have += n;
        }
        if have == 0 {
            return hist;
        }
        let mut part = [0u64; 256];
        with_ctx(|ctx| {
            let rc = unsafe { sys::hf_histogram_host(ctx, piece.as_ptr(), have as u64, part.as_mut_ptr()) };
            check(ctx, rc, "generate_histogram");
        });
        for (h, p) in hist.iter_mut().zip(part.iter()) {
            *h += *p as usize;
        }
        if have < piece.len() {
            return hist;
        }
    }
}

use rand::distributions::WeightedIndex;
use rand::prelude::*;

/// `count` bytes from `min..max` drawn with `weights` (reference: stats.rs:36-42;
/// unseeded there too, so no test data of the crate is reproducible).
pub fn generate_random_byte_vector(min: u8, max: u8, count: usize, weights: &[u8]) -> Vec<u8> {
    let choices: Vec<u8> = (min..max).collect();
    assert_eq!(choices.len(), weights.len());
    let dist = WeightedIndex::new(weights).unwrap();
    let mut rng = thread_rng();
    (0..count).map(|_| choices[dist.sample(&mut rng)]).collect()
}
