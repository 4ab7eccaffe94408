//! One stream over several GPUs, one process per GPU (no counterpart in the
//! reference, which encodes in one thread).  The histogram exchange itself is
//! the caller's collective (RCCL all-gather of 256 counts per rank); this is
//! the arithmetic around it and the assembly of the stream.
use hfgpu_sys as sys;

pub struct ShardPlan {
    pub bits: Vec<u64>,
    pub start_bit: Vec<u64>,
    pub lead_bits: Vec<u32>,
    pub byte_first: Vec<u64>,
    pub byte_last: Vec<u64>,
    pub total_bits: u64,
}

/// `hists` = the gathered histograms, rank r's 256 counts at r * 256; `lens` =
/// code lengths of the codebook built from their sum.  Rank r then encodes its
/// shard with `lead_bits[r]` and its stream covers bytes
/// `byte_first[r]..byte_last[r]` of the whole stream.
pub fn shard_plan(hists: &[u64], lens: &[u64; 256]) -> ShardPlan {
    assert!(!hists.is_empty() && hists.len() % 256 == 0);
    let world = hists.len() / 256;
    let mut p = ShardPlan {
        bits: vec![0; world], start_bit: vec![0; world], lead_bits: vec![0; world],
        byte_first: vec![0; world], byte_last: vec![0; world], total_bits: 0,
    };
    let rc = unsafe {
        sys::hf_shard_plan(hists.as_ptr(), lens.as_ptr(), world as u32, p.bits.as_mut_ptr(),
                           p.start_bit.as_mut_ptr(), p.lead_bits.as_mut_ptr(),
                           p.byte_first.as_mut_ptr(), p.byte_last.as_mut_ptr(), &mut p.total_bits)
    };
    assert_eq!(rc, sys::HF_OK);
    p
}

/// Puts a shard's local stream (lead bits included) at its place in `whole`
/// (zero initialised); boundary bytes are OR-merged, any order works.
pub fn place_shard(whole: &mut [u8], shard: &[u8], start_bit: u64, shard_bits: u64) {
    let rc = unsafe { sys::hf_place_shard(whole.as_mut_ptr(), whole.len() as u64, shard.as_ptr(), start_bit, shard_bits) };
    assert_eq!(rc, sys::HF_OK, "hf_place_shard");
}
