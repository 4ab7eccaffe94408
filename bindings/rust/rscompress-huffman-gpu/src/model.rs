//! The boundary type between codebook and codecs (reference: src/model.rs:3-7).
//! Unchanged by the GPU path: the device tables are built from what `encode`
//! returns for the 256 symbols.
use std::collections::BTreeMap;

pub trait Model {
    /// (codeword, length in bits) of `sym`; (0, 0) for a symbol without a code.
    fn encode(&self, sym: u8) -> (usize, usize);
    /// Longest code length of the model: the width of the decoder's bit window.
    fn sentinel(&self) -> usize;
    /// `codeword << (sentinel - length) -> (symbol, length)` for every coded symbol.
    fn to_btreemap(&self) -> BTreeMap<usize, (u8, u8)>;
}
