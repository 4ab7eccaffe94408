// Command line tool with the argument contract of the reference's binary
// (src/main.rs:17-38):   hfgpu <source> <destination> <method> <operation>
//   method "h":  operation "c" compresses into the "pzhf" container,
//                operation "d" decompresses it (the reference leaves this
//                `unimplemented!()`; here it is the inverse of "c");
//   any other method: header-less stream of the whole file (src/main.rs:40-97).
// Missing arguments end the process with status 101, as a Rust panic does.
// RUST_LOG=info (or debug/trace) prints the reference's progress lines.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "hfgpu.hpp"

using namespace rscompress_huffman;

static bool g_info = false;
static void info(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static void info(const char *fmt, ...)
{
    if (!g_info)
        return;
    va_list ap;
    va_start(ap, fmt);
    std::fputs("[INFO  hfgpu] ", stderr);
    std::vfprintf(stderr, fmt, ap);
    std::fputc('\n', stderr);
    va_end(ap);
}

[[noreturn]] static void panic(const char *msg)
{
    std::fprintf(stderr, "thread 'main' panicked: %s\n", msg);
    std::exit(101);
}

// src/main.rs:40-97
static void old_main(const std::string &source, const std::string &destination)
{
    info("Starting compression");
    info("Input:  %s", source.c_str());
    info("Output: %s", destination.c_str());
    FileReader reader(source);
    std::vector<u8> buffer(BUF);
    info("Generating histogram...");
    auto histogram = stats::generate_histogram(reader);
    info("Generating codewords...");
    auto cl = huffman::generate_extended_codewords(
        std::vector<usize>(histogram.begin(), histogram.end()));
    huffman::Huffman h(cl.first, cl.second);
    huffman::encode::Encoder<FileWriter> writer(FileWriter(destination), h);
    writer.track_write_value = false;
    if (g_info) { // src/main.rs:61-79
        usize original_file_size = 0, huffmann_file_size = 0;
        for (int i = 0; i < 256; ++i) {
            original_file_size += histogram[i];
            huffmann_file_size += histogram[i] * huffman::encode::calculate_length(cl.first[i]);
        }
        huffmann_file_size = huffmann_file_size / 8 + 1;
        info("Original file size: %llu", (unsigned long long)original_file_size);
        info("Huffman file size:  %llu", (unsigned long long)huffmann_file_size);
        info("Compression factor: %.2f", (float)original_file_size / (float)huffmann_file_size);
        info("Compression ratio:  %.2f", (float)huffmann_file_size / (float)original_file_size);
    }
    reader.rewind();
    info("Starting writing...");
    while (true) {
        const std::size_t n = reader.read(buffer.data(), buffer.size());
        if (n == 0)
            break;
        writer.write(buffer.data(), n);
    }
    writer.flush();
    write_gap_sidecar(gap_sidecar_path(destination), writer.gap());
    info("End compression");
}

int main(int argc, char **argv)
{
    if (const char *lv = std::getenv("RUST_LOG")) {
        const std::string s(lv);
        g_info = s.find("info") != std::string::npos || s.find("debug") != std::string::npos ||
                 s.find("trace") != std::string::npos;
    }
    if (argc < 2)
        panic("No source file found!");
    if (argc < 3)
        panic("No destination file found");
    if (argc < 4)
        panic("No method found");
    if (argc < 5)
        panic("No operation defined");
    const std::string source = argv[1], destination = argv[2], method = argv[3], operation = argv[4];
    try {
        if (method == "h") {
            info("Huffman with header information");
            if (operation == "c") {
                info("Compressing file ... ");
                stream_compress_with_header_information(source, destination);
            } else if (operation == "d") {
                info("Decompressing file ... ");
                stream_decompress_with_header_information(source, destination);
            } else {
                panic("Do not understand operation");
            }
        } else {
            info("Huffman without header information");
            old_main(source, destination);
        }
    } catch (const std::exception &e) {
        panic(e.what());
    }
    return 0;
}
