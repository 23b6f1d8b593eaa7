// tensor_archive.hpp -- reader for the native model_dir files (cae.bin, detector.bin).
// Format (little endian), written by cellscreen/model_io.py:
//   char magic[8] = "CSTENS01"; u32 count;
//   count x { u32 name_len; char name[name_len]; u32 dtype; u32 ndim; u64 dims[ndim];
//             u64 nbytes; pad to 8-byte file offset; u8 data[nbytes]; pad to 8 }
//   dtype: 0 = f32, 1 = f64, 2 = i32, 3 = i64
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace cs {

struct Tensor {
    uint32_t dtype = 0;
    std::vector<uint64_t> dims;
    std::vector<uint8_t> data;
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= d; return n; }
    const float* f32() const { return dtype == 0 ? (const float*)data.data() : nullptr; }
    const double* f64() const { return dtype == 1 ? (const double*)data.data() : nullptr; }
    const int32_t* i32() const { return dtype == 2 ? (const int32_t*)data.data() : nullptr; }
};

struct TensorArchive {
    std::map<std::string, Tensor> tensors;

    // returns empty string on success, else an error message
    std::string load(const std::string& path)
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return "cannot open " + path;
        std::string err;
        auto rd = [&](void* p, size_t n) { return fread(p, 1, n, f) == n; };
        auto align8 = [&]() { long pos = ftell(f); long pad = (8 - (pos & 7)) & 7; if (pad) fseek(f, pad, SEEK_CUR); };
        char magic[8];
        uint32_t count = 0;
        if (!rd(magic, 8) || memcmp(magic, "CSTENS01", 8) != 0) { fclose(f); return "bad magic in " + path; }
        if (!rd(&count, 4) || count > 100000) { fclose(f); return "bad tensor count in " + path; }
        static const size_t esize[4] = {4, 8, 4, 8};
        for (uint32_t i = 0; i < count; ++i) {
            uint32_t nl = 0;
            if (!rd(&nl, 4) || nl == 0 || nl > 256) { err = "bad name length"; break; }
            std::string name(nl, '\0');
            if (!rd(&name[0], nl)) { err = "truncated name"; break; }
            Tensor t;
            uint32_t nd = 0;
            if (!rd(&t.dtype, 4) || !rd(&nd, 4) || t.dtype > 3 || nd > 8) { err = "bad header for " + name; break; }
            t.dims.resize(nd);
            if (nd && !rd(t.dims.data(), 8 * nd)) { err = "truncated dims for " + name; break; }
            uint64_t nbytes = 0;
            if (!rd(&nbytes, 8)) { err = "truncated size for " + name; break; }
            if (nbytes != t.numel() * esize[t.dtype] || nbytes > (1ull << 34)) { err = "size mismatch for " + name; break; }
            align8();
            t.data.resize(nbytes);
            if (nbytes && !rd(t.data.data(), nbytes)) { err = "truncated data for " + name; break; }
            align8();
            tensors.emplace(std::move(name), std::move(t));
        }
        fclose(f);
        return err.empty() ? err : err + " in " + path;
    }

    const Tensor* get(const std::string& name) const
    {
        auto it = tensors.find(name);
        return it == tensors.end() ? nullptr : &it->second;
    }
};

}  // namespace cs
