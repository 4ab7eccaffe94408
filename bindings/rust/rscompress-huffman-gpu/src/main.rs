//! Command line of the crate (reference: src/main.rs:17-97):
//! `<source> <destination> <method> <operation>`; method "h" = container with
//! header (operation "c" or "d"), anything else = headerless stream.
use log::info;
use rscompress_huffman::huffman::encode::Encoder;
use rscompress_huffman::huffman::Huffman;
use rscompress_huffman::{stream_compress_with_header_information, stream_decompress_with_header_information, BUF};
use std::fs::File;
use std::io::{BufReader, BufWriter, Read, Write};

fn main() {
    env_logger::init();
    let args: Vec<String> = std::env::args().collect();
    let source = args.get(1).expect("No source file given");
    let destination = args.get(2).expect("No destination file given");
    let method = args.get(3).expect("No method given");
    let operation = args.get(4).expect("No operation given");
    if method == "h" {
        info!("Huffman with header information");
        match operation.as_str() {
            "c" => {
                info!("Compressing file ... ");
                stream_compress_with_header_information(source, destination)
            }
            "d" => {
                info!("Decompressing file ... ");
                stream_decompress_with_header_information(source, destination)
            }
            _ => panic!("Do not understand operation"),
        }
    } else {
        info!("Huffman without header information");
        old_main(source, destination);
    }
}

/// Headerless path: the bare bitstream of the whole file.
fn old_main(source: &str, destination: &str) {
    let mut reader = BufReader::new(File::open(source).expect("Failed to open source file"));
    let model = Huffman::from_reader(&mut reader);
    let out = BufWriter::new(File::create(destination).expect("Failed to create destination file"));
    let mut enc = Encoder::new(out, &model);
    enc.track_write_value = false;
    let mut buffer = vec![0u8; BUF];
    loop {
        let n = reader.read(&mut buffer).expect("Failed to read source");
        if n == 0 {
            break;
        }
        enc.write_all(&buffer[..n]).expect("Failed to encode");
    }
    enc.flush().expect("Failed to flush");
    info!("{} bytes in, {} bytes out", enc.readbytes, enc.writeout);
}
