// TEST HARNESS ONLY -- not part of the product and never loaded by it.
// Compiles the device line parser (pantas_amd/csrc/gaf_device.h) for the CPU under ASan/UBSan and runs it the
// way tokenize.hip does (line index, measure pass, offsets, write pass), then compares every array and counter
// with what the host tokeniser (gaf.cpp) makes of the same text.
//
// usage: hostsim_tokenize <gaf> <gfa> [thr]     exit 0 and "OK ..." when both agree, 1 and the first difference otherwise
//        hostsim_tokenize --dv                  self-check of the decimal dv > 0.1 rule against strtod
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include "gaf.h"
#include "gaf_device.h"

using namespace pantas;

struct WriteOut {
  uint32_t* nodes;
  uint32_t* ops;
  void node(uint32_t i, uint32_t idx) { nodes[i] = idx; }
  void op(uint32_t i, uint32_t w) { ops[i] = w; }
  void ops_done(uint32_t) {}
  void nodes_done(uint32_t) {}
};

static const char* kind_of(uint32_t te) {
  switch (te) {
    case TE_COLUMNS: return "IndexError";
    case TE_MAPQ: case TE_POSITIONS: case TE_NO_DV: return "ValueError";
    case TE_OP_LEN: case TE_OFFSETS: case TE_LONE_CR: return "Unsupported";
    case TE_NO_NODE: return "AssertionError";
    case TE_UNKNOWN_NODE: return "KeyError";
  }
  return "?";
}

static int dv_selfcheck() {
  std::mt19937_64 rng(7);
  const std::string mid = "100000000000000012490009027033011079765856266021728515625";
  std::vector<std::string> cases = {"0", "0.1", "0.10", "0.1000000000000000", "0.09999999999999999999", "1", "00.1", "0.2", "0." + mid,
                                    "0." + mid + "0", "0." + mid + "1", "0." + mid.substr(0, 56) + "4", "0." + mid.substr(0, 56) + "6",
                                    "0.10000000000000001", "0.100000000000000012", "0.100000000000000013", "10", "0.0", "000", "0.1000000000000000125"};
  for (int i = 0; i < 200000; ++i) {
    std::string s;
    int kind = int(rng() % 4);
    if (kind == 0) { s = "0."; int n = 1 + int(rng() % 60); for (int k = 0; k < n; ++k) s.push_back(char('0' + rng() % 10)); }
    else if (kind == 1) { s = "0." + mid.substr(0, rng() % 58); int n = int(rng() % 4); for (int k = 0; k < n; ++k) s.push_back(char('0' + rng() % 10)); if (s == "0.") s = "0.1"; }
    else if (kind == 2) { s = "0.1"; int n = int(rng() % 30); for (int k = 0; k < n; ++k) s.push_back('0'); s.push_back(char('0' + rng() % 10)); }
    else { s = std::to_string(rng() % 3); if (rng() % 2) { s += "."; s += std::to_string(rng() % 1000); } }
    cases.push_back(s);
  }
  for (const std::string& s : cases) {
    const bool want = strtod(s.c_str(), nullptr) > 0.1;
    std::vector<uint32_t> words(s.size() / 4 + 2, 0u);       // (the parser reads its text through aligned words)
    memcpy(words.data(), s.data(), s.size());
    TextWords T(words.data());
    const bool got = tk_dv_above_tenth(T, 0, uint32_t(s.size()));
    if (want != got) { printf("DV MISMATCH %s want %d got %d\n", s.c_str(), int(want), int(got)); return 1; }
  }
  printf("OK dv cases=%zu\n", cases.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "--dv")) return dv_selfcheck();
  if (argc < 3) { fprintf(stderr, "usage: hostsim_tokenize <gaf> <gfa> [thr]\n"); return 2; }
  const int thr = argc > 3 ? atoi(argv[3]) : 20;
  try {
    auto g = load_graph(argv[2], 1);
    if (!g->names.dense) { printf("OK skipped (segment names are not canonical decimals)\n"); return 0; }
    MappedFile f;
    f.open(argv[1]);
    GafBatch host;
    tokenize_gaf(*g, f.data, f.size, thr, 0, argc > 4 ? unsigned(atoi(argv[4])) : 1u, host);

    // ---- the device pipeline, serially
    // (the device parser reads the text through aligned 32-bit words: up to three readable bytes behind the last line)
    std::vector<uint32_t> padded((f.size + 3) / 4 + 2, 0u);
    memcpy(padded.data(), f.data, f.size);
    const uint8_t* text = reinterpret_cast<const uint8_t*>(padded.data());
    std::vector<uint64_t> ls;   // line i = [ls[i], ls[i+1] - 1)
    ls.push_back(0);
    for (size_t i = 0; i < f.size; ++i) if (text[i] == '\n') ls.push_back(i + 1);
    if (f.size && text[f.size - 1] != '\n') ls.push_back(f.size + 1);
    const size_t n_lines = ls.size() - 1;
    DenseNames names{g->names.table.data(), g->names.base, g->names.table.size()};
    std::vector<LineSummary> sum(n_lines);
    size_t err_line = n_lines;
    // the byte-at-a-time form of the parser on the same line: what the word-at-a-time form must reproduce
    auto plain = [&](size_t i, auto& sink, LineSummary& out_sum) {
      const uint8_t* lb = text + ls[i];
      const uintptr_t a = reinterpret_cast<uintptr_t>(lb) & ~uintptr_t(3);
      TextWords T(reinterpret_cast<const uint32_t*>(a));
      const uint32_t off = uint32_t(reinterpret_cast<uintptr_t>(lb) - a);
      parse_gaf_line_bytewise(T, off, off + uint32_t(ls[i + 1] - 1 - ls[i]), thr, names, sink, out_sum);
    };
    for (size_t i = 0; i < n_lines; ++i) {
      CountOnly co;
      parse_gaf_line(text + ls[i], text + ls[i + 1] - 1, thr, names, co, sum[i]);
      if (sum[i].status == TOK_ERROR && err_line == n_lines) err_line = i;
      LineSummary ref;
      CountOnly co2;
      plain(i, co2, ref);
      if (memcmp(&ref, &sum[i], sizeof ref) != 0) {
        printf("MISMATCH line %zu: word-at-a-time parse (status %u err %u nodes %u ops %u flags %u start %d tail %d) differs from the plain one (%u %u %u %u %u %d %d)\n", i + 1,
               sum[i].status, sum[i].err, sum[i].n_nodes, sum[i].n_ops, sum[i].flags, sum[i].start, sum[i].tail, ref.status, ref.err, ref.n_nodes, ref.n_ops, ref.flags, ref.start, ref.tail);
        return 1;
      }
    }
    // ---- the fast form for plain lines (parse_gaf_line_fast): whatever it accepts, it decides and emits as the general parser does
    size_t fast_taken = 0;
    bool lone_cr = false;         // what tokenize.hip's line index reports for the text: a '\r' that no '\n' follows
    for (size_t i = 0; i < f.size; ++i) lone_cr = lone_cr || (text[i] == '\r' && !(i + 1 < f.size && text[i + 1] == '\n'));
    for (size_t i = 0; i < n_lines; ++i) {
      const uint8_t* lb = text + ls[i];
      const uintptr_t a = reinterpret_cast<uintptr_t>(lb) & ~uintptr_t(3);
      const uint32_t off = uint32_t(reinterpret_cast<uintptr_t>(lb) - a), len = uint32_t(ls[i + 1] - 1 - ls[i]);
      std::vector<uint32_t> fn(len + 2, 0xDEADBEEFu), fo(len + 2, 0xDEADBEEFu), gn(len + 2, 0xDEADBEEFu), go(len + 2, 0xDEADBEEFu);
      WriteOut fw{fn.data(), fo.data()}, gw{gn.data(), go.data()};
      LineSummary fs, gs;
      TextWords TF(reinterpret_cast<const uint32_t*>(a));
      if (!parse_gaf_line_fast(TF, off, off + len, thr, names, fw, fs, lone_cr)) continue;
      ++fast_taken;
      TextWords TG(reinterpret_cast<const uint32_t*>(a));
      parse_gaf_line_words(TG, off, off + len, thr, names, gw, gs);
      if (memcmp(&fs, &gs, sizeof fs) != 0) {
        printf("MISMATCH line %zu: fast parse (status %u err %u nodes %u ops %u flags %u start %d tail %d) differs from the general one (%u %u %u %u %u %d %d)\n", i + 1,
               fs.status, fs.err, fs.n_nodes, fs.n_ops, fs.flags, fs.start, fs.tail, gs.status, gs.err, gs.n_nodes, gs.n_ops, gs.flags, gs.start, gs.tail);
        return 1;
      }
      if (fs.status == TOK_KEEP && (memcmp(fn.data(), gn.data(), (fs.n_nodes + 1) * 4) != 0 || memcmp(fo.data(), go.data(), (fs.n_ops + 1) * 4) != 0)) {
        printf("MISMATCH line %zu: fast parse emits other nodes / ops than the general one\n", i + 1);
        return 1;
      }
    }
    GafBatch dev;
    dev.node_off.push_back(0);
    dev.op_off.push_back(0);
    for (size_t i = 0; i < err_line; ++i) {
      const LineSummary& s = sum[i];
      if (s.status == TOK_REJECT) ++dev.rejected;
      if (s.status == TOK_SKIP) ++dev.skipped;
      if (s.status != TOK_KEEP) continue;
      const size_t nb = dev.nodes.size(), ob = dev.ops.size();
      // guard words behind each record catch a write pass that emits more than the measure pass counted
      dev.nodes.resize(nb + s.n_nodes + 1, 0xDEADBEEFu);
      dev.ops.resize(ob + s.n_ops + 1, 0xDEADBEEFu);
      WriteOut wo{&dev.nodes[nb], &dev.ops[ob]};
      LineSummary again;
      parse_gaf_line(text + ls[i], text + ls[i + 1] - 1, thr, names, wo, again);
      if (memcmp(&again, &s, sizeof s) != 0) { printf("MISMATCH line %zu: write pass summary differs from measure pass\n", i + 1); return 1; }
      if (dev.nodes.back() != 0xDEADBEEFu || dev.ops.back() != 0xDEADBEEFu) { printf("MISMATCH line %zu: write pass overran its slots\n", i + 1); return 1; }
      {
        std::vector<uint32_t> pn(s.n_nodes + 1, 0xDEADBEEFu), po(s.n_ops + 1, 0xDEADBEEFu);
        WriteOut pw{pn.data(), po.data()};
        LineSummary ref;
        plain(i, pw, ref);
        if (memcmp(pn.data(), &dev.nodes[nb], (s.n_nodes + 1) * 4) != 0 || memcmp(po.data(), &dev.ops[ob], (s.n_ops + 1) * 4) != 0) {
          printf("MISMATCH line %zu: word-at-a-time parse emits other nodes / ops than the plain one\n", i + 1);
          return 1;
        }
      }
      dev.nodes.pop_back();
      dev.ops.pop_back();
      dev.hdr.push_back(LineHdr{s.start, s.tail, uint32_t(i), s.flags});
      dev.node_off.push_back(uint32_t(dev.nodes.size()));
      dev.op_off.push_back(uint32_t(dev.ops.size()));
      if (s.flags & LINE_HAS_DEL) { ++dev.del_lines; dev.del_nodes += s.n_nodes; }
    }
    dev.lines_total = err_line < n_lines ? err_line + 1 : n_lines;

    // ---- compare
    const bool host_err = bool(host.deferred);
    if (host_err != (err_line < n_lines)) { printf("MISMATCH error presence: host %d device %d (device line %zu)\n", int(host_err), int(err_line < n_lines), err_line + 1); return 1; }
    if (host_err) {
      const std::string want_kind = host.deferred->kind;
      const std::string msg = host.deferred->what();
      const std::string tag = "GAF line " + std::to_string(err_line + 1) + ":";
      if (want_kind != kind_of(sum[err_line].err) || msg.find(tag) == std::string::npos) {
        printf("MISMATCH error: host %s '%s' device %s at line %zu\n", want_kind.c_str(), msg.c_str(), kind_of(sum[err_line].err), err_line + 1);
        return 1;
      }
    }
#define SAME(field) if (host.field != dev.field) { printf("MISMATCH " #field ": host %llu device %llu\n", (unsigned long long)host.field, (unsigned long long)dev.field); return 1; }
    SAME(lines_total) SAME(rejected) SAME(skipped) SAME(del_lines) SAME(del_nodes)
    if (host.size() != dev.size()) { printf("MISMATCH records: host %zu device %zu\n", host.size(), dev.size()); return 1; }
    for (size_t i = 0; i < host.size(); ++i) {
      const LineHdr &a = host.hdr[i], &b = dev.hdr[i];
      if (a.start != b.start || a.tail != b.tail || a.line_id != b.line_id || a.flags != b.flags) {
        printf("MISMATCH hdr of record %zu (line %u): host {%d,%d,%u,%u} device {%d,%d,%u,%u}\n", i, a.line_id + 1, a.start, a.tail, a.line_id, a.flags, b.start, b.tail, b.line_id, b.flags);
        return 1;
      }
    }
    if (host.node_off != dev.node_off || host.nodes != dev.nodes) { printf("MISMATCH path nodes\n"); return 1; }
    if (host.op_off != dev.op_off || host.ops != dev.ops) { printf("MISMATCH cs ops\n"); return 1; }
    printf("OK lines=%zu records=%zu rejected=%llu skipped=%llu error=%s fast=%zu\n", n_lines, dev.size(), (unsigned long long)dev.rejected,
           (unsigned long long)dev.skipped, host_err ? host.deferred->kind.c_str() : "none", fast_taken);
    return 0;
  } catch (const Error& e) {
    // the graph itself is refused (both tokenisers are behind the same loader)
    printf("OK skipped (%s)\n", e.what());
    return 0;
  }
}
