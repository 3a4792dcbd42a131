// FileIO.hxx -- the file object decks write through (the reference's src/util/io/FileIO.hxx:22-62
// over its StandardIOPolicy): open / close / print / typed read and write / seek.  Here a thin
// shell around stdio; same member names, argument meaning and status values.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
// vpic_hip_host.cxx: a host mirror must be faulted in (and, before the kernel writes into it, marked dirty)
// before a system call works on it
void vpic_host_touch(const void *p, size_t bytes);
void vpic_host_touch_for_write(void *p, size_t bytes);

enum FileIOStatus { fail = 0, ok = 1 };
enum FileIOMode { io_read, io_read_write, io_write, io_write_read, io_append, io_append_read };

class FileIO {
  FILE *f_;
public:
  FileIO() : f_(NULL) {}
  ~FileIO() {}                       // like the reference: a file left open is not closed behind the deck's back
  FileIOStatus open(const char *filename, FileIOMode mode) {
    static const char *how[] = {"r", "r+", "w", "w+", "a", "a+"};
    f_ = fopen(filename, how[mode]);
    return f_ ? ok : fail;
  }
  void close() { if (f_) fclose(f_); f_ = NULL; }
  bool isOpen() { return f_ != NULL; }
  int64_t size() {
    const long here = ftell(f_);
    fseek(f_, 0L, SEEK_END);
    const long n = ftell(f_);
    fseek(f_, here, SEEK_SET);
    return n;
  }
  void print(const char *format, ...) {
    va_list args;
    va_start(args, format);
    vfprintf(f_, format, args);
    va_end(args);
  }
  template <typename T> void read(T *data, size_t elements) { vpic_host_touch_for_write(data, sizeof(T) * elements); if (fread(data, sizeof(T), elements, f_) != elements) {} }
  template <typename T> void write(const T *data, size_t elements) { vpic_host_touch(data, sizeof(T) * elements); fwrite(data, sizeof(T), elements, f_); }
  void seek(uint64_t offset, int32_t whence) { fseek(f_, (long)offset, whence); }
  int64_t tell() { return ftell(f_); }
  void rewind() { ::rewind(f_); }
  void flush() { fflush(f_); }
};
typedef FileIO FileIOUnswapped;
