// state_io.hpp -- per-rank state dump: the role of save_model_state / save_sharded_fields of GB-25
// src/sharded_io.jl:70-96,122-138 (called after each loop of the benchmark script,
// sharding/sharded_baroclinic_instability_simulation_run.jl:151-155,167-171).  Every rank writes ONLY its own slab, no
// communication: <dir>/<label>/fields_rank<R>.npz holding, per field of Oceananigans.fields(model) = (u, v, w, eta, T,
// S): the local interior array (`<name>.data`), its slice in the global array (`<name>.slice` = i0, i1, j0, j1, k0, k1,
// 0-based half-open) and the global shape (`<name>.global_shape`), plus iteration, time, rank, nranks.  The container
// is an uncompressed .npz (a ZIP archive of NumPy .npy members, fortran_order so that index order is [i, j, k]):
// np.load reads it, so does NPZ.jl on the Julia side.  Offline re-assembly: gb-25_amd/sharded_io.py (load_all_fields).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <sys/stat.h>

namespace {

struct NpzWriter {
  FILE* f = nullptr;
  struct Entry {
    std::string name;
    uint32_t crc, size, offset;
  };
  std::vector<Entry> entries;
  bool ok = true;

  static uint32_t crc32(const unsigned char* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[8][256];
    static bool init = false;
    if (!init) {
      for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        table[0][i] = c;
      }
      for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) table[t][i] = table[0][table[t - 1][i] & 255] ^ (table[t - 1][i] >> 8);
      init = true;
    }
    crc = ~crc;
    while (n >= 8) {   // slice-by-8
      uint32_t a = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
      a ^= crc;
      crc = table[7][a & 255] ^ table[6][(a >> 8) & 255] ^ table[5][(a >> 16) & 255] ^ table[4][a >> 24] ^
            table[3][p[4]] ^ table[2][p[5]] ^ table[1][p[6]] ^ table[0][p[7]];
      p += 8;
      n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 255] ^ (crc >> 8);
    return ~crc;
  }
  void w16(uint16_t v) { ok = ok && fwrite(&v, 2, 1, f) == 1; }
  void w32(uint32_t v) { ok = ok && fwrite(&v, 4, 1, f) == 1; }
  bool open(const std::string& path) {
    f = fopen(path.c_str(), "wb");
    return f != nullptr;
  }
  // one stored (uncompressed) member made of a header part and a data part
  void add(const std::string& name, const std::string& head, const void* data, size_t nbytes) {
    const uint64_t total = head.size() + nbytes;
    if (total >= 0xFFFFFFFFull || (uint64_t)ftell(f) >= 0xFFFFFFFFull) {   // (zip32; a slab's field stays far below)
      ok = false;
      return;
    }
    uint32_t crc = crc32((const unsigned char*)head.data(), head.size());
    crc = crc32((const unsigned char*)data, nbytes, crc);
    Entry e{name, crc, (uint32_t)total, (uint32_t)ftell(f)};
    w32(0x04034b50); w16(20); w16(0); w16(0); w16(0); w16(0x21);   // local header: stored, dos time 0, date 1980-01-01
    w32(e.crc); w32(e.size); w32(e.size); w16((uint16_t)name.size()); w16(0);
    ok = ok && fwrite(name.data(), 1, name.size(), f) == name.size();
    ok = ok && fwrite(head.data(), 1, head.size(), f) == head.size();
    if (nbytes) ok = ok && fwrite(data, 1, nbytes, f) == nbytes;
    entries.push_back(e);
  }
  static std::string npy_header(const char* descr, const std::vector<long>& shape) {
    std::string sh;
    for (size_t q = 0; q < shape.size(); q++) sh += std::to_string(shape[q]) + (shape.size() == 1 || q + 1 < shape.size() ? ", " : "");
    std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': True, 'shape': (" + sh + "), }";
    size_t len = 10 + dict.size() + 1;
    dict.append((64 - len % 64) % 64, ' ');
    dict += '\n';
    std::string out("\x93NUMPY\x01\x00", 8);
    out += (char)(dict.size() & 255);
    out += (char)(dict.size() >> 8);
    return out + dict;
  }
  template <class T>
  void add_array(const std::string& name, const char* descr, const std::vector<long>& shape, const T* data) {
    size_t n = 1;
    for (long s : shape) n *= (size_t)s;
    add(name + ".npy", npy_header(descr, shape), data, n * sizeof(T));
  }
  bool close() {
    if (!f) return false;
    const uint32_t cd_off = (uint32_t)ftell(f);
    for (auto& e : entries) {
      w32(0x02014b50); w16(20); w16(20); w16(0); w16(0); w16(0); w16(0x21);
      w32(e.crc); w32(e.size); w32(e.size); w16((uint16_t)e.name.size()); w16(0); w16(0); w16(0); w16(0); w32(0);
      w32(e.offset);
      ok = ok && fwrite(e.name.data(), 1, e.name.size(), f) == e.name.size();
    }
    const uint32_t cd_size = (uint32_t)ftell(f) - cd_off;
    w32(0x06054b50); w16(0); w16(0); w16((uint16_t)entries.size()); w16((uint16_t)entries.size());
    w32(cd_size); w32(cd_off); w16(0);
    ok = (fclose(f) == 0) && ok;
    f = nullptr;
    return ok;
  }
};

}  // namespace
