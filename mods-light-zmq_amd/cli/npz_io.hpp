// .npz keypoint files: what cnpy::npz_save / npz_load do for the reference (imagerepresentation.cpp:1257-1316, 1355-1513) -
// a ZIP archive of uncompressed ("stored") .npy members, NumPy format 1.0, C order, little endian.
// Members written by SaveRegionsNPZ: xy (n,2) f8 | scales (n,1) f8 | responses (n,1) f8 | A (n,4) f8 | descs (n,dim) u1.
// The reader accepts stored members only (np.savez, cnpy); deflated archives (np.savez_compressed) are refused.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace modscli {

struct NpyArray {
  std::string descr;                 // "<f8", "|u1", ...
  std::vector<size_t> shape;
  std::vector<unsigned char> data;   // raw, C order
  size_t count() const { size_t n = 1; for (size_t s : shape) n *= s; return n; }
};

inline uint32_t crc32_of(const unsigned char *p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

inline void put16(std::vector<unsigned char> &v, uint16_t x) { v.push_back(x & 0xFF); v.push_back(x >> 8); }
inline void put32(std::vector<unsigned char> &v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((x >> (8 * i)) & 0xFF); }

inline std::vector<unsigned char> npy_bytes(const NpyArray &a) {
  std::string dict = "{'descr': '" + a.descr + "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < a.shape.size(); i++) {
    dict += std::to_string(a.shape[i]);
    if (i + 1 < a.shape.size()) dict += ", ";
  }
  if (a.shape.size() == 1) dict += ",";
  dict += "), }";
  size_t total = 10 + dict.size() + 1;               // magic(6) version(2) len(2) dict '\n'
  const size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict += '\n';
  std::vector<unsigned char> out;
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  out.insert(out.end(), magic, magic + 8);
  put16(out, (uint16_t)dict.size());
  out.insert(out.end(), dict.begin(), dict.end());
  out.insert(out.end(), a.data.begin(), a.data.end());
  return out;
}

// members in the given order (the reference writes xy, scales, responses, A, descs)
inline bool npz_write(const std::string &fn, const std::vector<std::pair<std::string, NpyArray>> &members, std::string *err) {
  std::vector<unsigned char> file, central;
  for (const auto &m : members) {
    const std::string name = m.first + ".npy";
    const std::vector<unsigned char> body = npy_bytes(m.second);
    if (body.size() > 0xFFFFFFFFull || file.size() > 0xFFFFFFFFull) { if (err) *err = "npz member too large (no zip64)"; return false; }
    const uint32_t crc = crc32_of(body.data(), body.size()), off = (uint32_t)file.size();
    put32(file, 0x04034b50); put16(file, 20); put16(file, 0); put16(file, 0); put16(file, 0); put16(file, 0x21);
    put32(file, crc); put32(file, (uint32_t)body.size()); put32(file, (uint32_t)body.size());
    put16(file, (uint16_t)name.size()); put16(file, 0);
    file.insert(file.end(), name.begin(), name.end());
    file.insert(file.end(), body.begin(), body.end());
    put32(central, 0x02014b50); put16(central, 20); put16(central, 20); put16(central, 0); put16(central, 0); put16(central, 0); put16(central, 0x21);
    put32(central, crc); put32(central, (uint32_t)body.size()); put32(central, (uint32_t)body.size());
    put16(central, (uint16_t)name.size()); put16(central, 0); put16(central, 0); put16(central, 0); put16(central, 0); put32(central, 0);
    put32(central, off);
    central.insert(central.end(), name.begin(), name.end());
  }
  const uint32_t cd_off = (uint32_t)file.size(), cd_size = (uint32_t)central.size();
  file.insert(file.end(), central.begin(), central.end());
  put32(file, 0x06054b50); put16(file, 0); put16(file, 0); put16(file, (uint16_t)members.size()); put16(file, (uint16_t)members.size());
  put32(file, cd_size); put32(file, cd_off); put16(file, 0);
  FILE *f = fopen(fn.c_str(), "wb");
  if (!f) { if (err) *err = "cannot open " + fn; return false; }
  const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
  fclose(f);
  if (!ok && err) *err = "short write to " + fn;
  return ok;
}

inline uint16_t get16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t get32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

inline bool npy_parse(const unsigned char *p, size_t n, NpyArray *a, std::string *err) {
  if (n < 10 || p[0] != 0x93 || memcmp(p + 1, "NUMPY", 5) != 0) { if (err) *err = "not a .npy member"; return false; }
  size_t hlen, hoff;
  if (p[6] == 1) { hlen = get16(p + 8); hoff = 10; }
  else { if (n < 12) return false; hlen = get32(p + 8); hoff = 12; }
  if (hoff + hlen > n) { if (err) *err = "truncated .npy header"; return false; }
  const std::string h((const char *)p + hoff, hlen);
  // every field is looked up defensively: a corrupt or hostile k1 / k2 file must be refused, not read out of bounds
  size_t i = h.find("'descr'");
  if (i == std::string::npos) { if (err) *err = "no descr in .npy header"; return false; }
  const size_t colon = h.find(':', i);
  i = colon == std::string::npos ? colon : h.find('\'', colon);
  const size_t j = i == std::string::npos ? i : h.find('\'', i + 1);
  if (j == std::string::npos) { if (err) *err = "malformed descr in .npy header"; return false; }
  a->descr = h.substr(i + 1, j - i - 1);
  if (h.find("'fortran_order': True") != std::string::npos) { if (err) *err = "Fortran-order arrays are not supported"; return false; }
  const size_t sh = h.find("'shape'");
  i = sh == std::string::npos ? sh : h.find('(', sh);
  const size_t e = i == std::string::npos ? i : h.find(')', i);
  if (e == std::string::npos) { if (err) *err = "malformed shape in .npy header"; return false; }
  a->shape.clear();
  size_t pos = i + 1;
  while (pos < e) {
    while (pos < e && (h[pos] == ' ' || h[pos] == ',')) pos++;
    if (pos >= e) break;
    if (h[pos] < '0' || h[pos] > '9') { if (err) *err = "malformed shape in .npy header"; return false; }
    size_t v = 0;
    while (pos < e && h[pos] >= '0' && h[pos] <= '9') {
      if (v > (SIZE_MAX - 9) / 10) { if (err) *err = ".npy dimension overflows"; return false; }
      v = v * 10 + (size_t)(h[pos++] - '0');
    }
    a->shape.push_back(v);
  }
  size_t item = 0;
  if (a->descr.size() >= 3) { const int it = atoi(a->descr.c_str() + 2); item = it > 0 && it <= 16 ? (size_t)it : 0; }
  if (item == 0) { if (err) *err = "bad .npy element type " + a->descr; return false; }
  const size_t avail = n - hoff - hlen;      // hoff + hlen <= n was checked above
  size_t bytes = item;
  for (size_t d : a->shape) {
    if (d != 0 && bytes > avail / d) { if (err) *err = "bad .npy payload size"; return false; }   // also rules out overflow
    bytes *= d;
  }
  if (bytes > avail) { if (err) *err = "bad .npy payload size"; return false; }
  a->data.assign(p + hoff + hlen, p + hoff + hlen + bytes);
  return true;
}

inline bool npz_read(const std::string &fn, std::map<std::string, NpyArray> *out, std::string *err) {
  FILE *f = fopen(fn.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + fn; return false; }
  std::vector<unsigned char> buf;
  unsigned char tmp[65536];
  size_t r;
  while ((r = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
  fclose(f);
  if (buf.size() < 22) { if (err) *err = fn + ": not a zip archive"; return false; }
  size_t eocd = std::string::npos;
  for (size_t i = buf.size() - 22 + 1; i-- > 0;) {
    if (get32(&buf[i]) == 0x06054b50) { eocd = i; break; }
    if (buf.size() - i > 65557) break;
  }
  if (eocd == std::string::npos) { if (err) *err = fn + ": no end-of-central-directory record"; return false; }
  const int n_entries = get16(&buf[eocd + 10]);
  size_t p = get32(&buf[eocd + 16]);
  for (int k = 0; k < n_entries; k++) {
    if (p + 46 > buf.size() || get32(&buf[p]) != 0x02014b50) { if (err) *err = fn + ": bad central directory"; return false; }
    const int method = get16(&buf[p + 10]);
    const size_t csize = get32(&buf[p + 20]), usize = get32(&buf[p + 24]);
    const size_t nlen = get16(&buf[p + 28]), xlen = get16(&buf[p + 30]), clen = get16(&buf[p + 32]);
    const size_t loff = get32(&buf[p + 42]);
    if (p + 46 + nlen + xlen + clen > buf.size()) { if (err) *err = fn + ": truncated central directory"; return false; }
    std::string name((const char *)&buf[p + 46], nlen);
    p += 46 + nlen + xlen + clen;
    if (method != 0 || csize != usize) { if (err) *err = fn + ": member " + name + " is compressed (only stored members are read)"; return false; }
    if (loff + 30 > buf.size() || get32(&buf[loff]) != 0x04034b50) { if (err) *err = fn + ": bad local header"; return false; }
    const size_t data = loff + 30 + get16(&buf[loff + 26]) + get16(&buf[loff + 28]);
    if (data > buf.size() || usize > buf.size() - data) { if (err) *err = fn + ": truncated member " + name; return false; }
    if (name.size() > 4 && name.substr(name.size() - 4) == ".npy") name.resize(name.size() - 4);
    NpyArray a;
    if (!npy_parse(&buf[data], usize, &a, err)) return false;
    (*out)[name] = std::move(a);
  }
  return true;
}

}  // namespace modscli
