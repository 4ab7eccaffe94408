//! Container header (reference: src/huffman/header.rs): serde + bincode, so the
//! bytes are whatever the crate's own dependency versions produce.
use bincode::{deserialize, serialize};
use serde::{Deserialize, Serialize};
use std::collections::BTreeMap;

#[derive(Serialize, Deserialize, Debug, PartialEq)]
pub struct Header {
    pub magic: Vec<u8>,
    pub btree: BTreeMap<usize, (u8, u8)>,
    pub sentinel: usize,
    pub readbytes: usize,
}

use super::encode::Encoder;
use crate::model::Model;
use std::convert::From;
use std::io::Write;

impl<'a, W: Write, M: Model> From<&Encoder<'a, W, M>> for Header {
    fn from(enc: &Encoder<'a, W, M>) -> Self {
        Header {
            magic: enc.magic(),
            btree: enc.model.to_btreemap(),
            sentinel: enc.model.sentinel(),
            readbytes: enc.readbytes,
        }
    }
}

impl Header {
    pub fn to_binary(&self) -> Vec<u8> {
        serialize(&self).unwrap()
    }
    pub fn from_binary(vec: &[u8]) -> Self {
        deserialize(vec).unwrap()
    }
    pub fn update_readbytes(&mut self, bytes: u64) {
        self.readbytes = bytes as usize;
    }
}
