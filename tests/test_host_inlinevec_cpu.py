"""InlineVec (tantivy_amd/host/searcher.hpp): the small vector a Weight keeps its term ids and weights in — inline up to
N elements, heap beyond; copy / move / assignment / growth / resize semantics against std::vector, compiled with g++
(no GPU, no HIP)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "searcher.hpp"
#include <cstdio>
#include <random>
#include <vector>
using tantivy_amd::InlineVec;
static int fails = 0;
template <typename V> static void same(const V &a, const std::vector<uint32_t> &b, const char *what) {
  bool ok = a.size() == b.size() && a.empty() == b.empty();
  for (size_t i = 0; ok && i < b.size(); ++i) ok = a[i] == b[i] && a.data()[i] == b[i];
  size_t n = 0;
  for (uint32_t x : a) ok = ok && n < b.size() && x == b[n++];
  if (!ok || n != b.size()) { std::printf("MISMATCH %s\n", what); ++fails; }
}
int main() {
  std::mt19937 rng(5);
  for (int round = 0; round < 2000; ++round) {
    InlineVec<uint32_t, 4> v;
    std::vector<uint32_t> ref;
    const int ops = 1 + rng() % 24;
    for (int o = 0; o < ops; ++o) {
      switch (rng() % 7) {
        case 0: case 1: { uint32_t x = rng(); v.push_back(x); ref.push_back(x); break; }
        case 2: { size_t n = rng() % 12; v.resize(n, 7u); ref.resize(n, 7u); break; }
        case 3: { std::vector<uint32_t> src(rng() % 10); for (auto &x : src) x = rng(); v.assign(src.data(), src.data() + src.size()); ref = src; break; }
        case 4: { v = {1u, 2u, 3u}; ref = {1u, 2u, 3u}; break; }
        case 5: { v.clear(); ref.clear(); break; }
        case 6: { InlineVec<uint32_t, 4> c(v); same(c, ref, "copy ctor"); InlineVec<uint32_t, 4> m(std::move(c)); same(m, ref, "move ctor");
                  InlineVec<uint32_t, 4> a; a.push_back(9u); a = m; same(a, ref, "copy assign"); InlineVec<uint32_t, 4> b; b.resize(9, 1u); b = std::move(a); same(b, ref, "move assign");
                  v = v; same(v, ref, "self assign"); break; }
      }
      same(v, ref, "after op");
    }
    // a vector of them (what prepare_into holds): growth moves the elements
    std::vector<InlineVec<uint32_t, 4>> many;
    for (int i = 0; i < 9; ++i) many.push_back(v);
    for (auto &x : many) same(x, ref, "in a std::vector");
  }
  std::printf(fails ? "FAILED %d\n" : "OK\n", fails);
  return fails ? 1 : 0;
}
'''


def test_inlinevec_behaves_like_a_vector(tmp_path):
    src = tmp_path / "iv.cpp"
    src.write_text(SRC)
    exe = tmp_path / "iv"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         "-I", os.path.join(ROOT, "tantivy_amd", "host"), "-I", os.path.join(ROOT, "include"),
                         str(src), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("no sanitizer runtime: " + cc.stderr[-200:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
