// launch_custom.hip -- user-supplied drifts: the reference takes any Python callable as the drift
// (ParamsCDNLGSSMDynamics.drift, cdnlgssm_utils.py:38-61); a callable cannot cross a C ABI, so the counterpart here is
// a drift given as C source for f(x, theta), its Jacobian and (optionally) grad(div f), compiled at run time with
// hipRTC into the SAME lane-per-trajectory sweep bodies (filter_reg_body / ekf_smoother_reg_body of
// cdkf_reg_kernels.h) the built-in drifts use.  One module per (drift, precision, emission_dim, algorithm variant),
// compiled on first use and cached for the life of the process.  Above six state or emission dimensions (up to what the workgroup
// kernels' LDS plan holds, <= 64) the same source is compiled into the workgroup-per-trajectory sweeps of cdkf_wg2_kernels.h instead
// (launch_custom_wg): Jacobian, grad(div f) and sigma-point evaluations spread over the workgroup's threads, all by dual numbers.
#include <atomic>
#include <dirent.h>
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "cdkf_launch.h"
#include "cdkf_ukf_tangent_kernels.h"
#include "cdkf_wg2_kernels.h"  // WgArgs: the argument block of the workgroup kernels (drifts above six state dimensions)

namespace cdkf {

namespace {

struct CustomDrift {
  int d, n_theta;
  std::string f_src, jac_src, g_src;
  bool has_g;
  bool auto_jac, auto_g;  // the Jacobian / grad(div f) are derived from f_src by dual numbers (cdkf_dual.h)
};
struct CustomEmission {
  int d, m;
  std::string h_src, jac_src;
};
std::vector<CustomDrift> g_drifts;
std::vector<CustomEmission> g_emis;  // emission_kind = CDKF_EMISSION_CUSTOM_BASE + index
std::mutex g_mutex;
std::string g_src_dir;

// kind, bytes per real, emission_dim, ukf, zeroth, forecast, smoother (2: the log-likelihood gradient sweep), generic Runge-Kutta
// tableau, emission kind (0: linear)
// kind, bytes per real, m, ukf, zeroth, forecast, smoother (0 filter, 1 smoother, 2 gradient), generic tableau, emission kind, input_dim
using Key = std::tuple<int, int, int, int, int, int, int, int, int, int>;
struct Compiled {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
};
std::map<std::pair<int, Key>, Compiled> g_modules;  // per device: a hipModule / hipFunction belongs to the device it was loaded on

std::mutex& g_mutex_emis() {
  static std::mutex m;
  return m;
}

std::mutex& g_mutex_srcdir() {
  static std::mutex m;
  return m;
}

bool blank(const std::string& t) { return t.find_first_not_of(" \t\r\n;") == std::string::npos; }

std::string source_dir() {
  std::lock_guard<std::mutex> lock(g_mutex_srcdir());
  if (!g_src_dir.empty()) return g_src_dir;
  Dl_info info;
  if (dladdr((const void*)&source_dir, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t slash = p.find_last_of('/');
    g_src_dir = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/../csrc";
  } else {
    g_src_dir = ".";
  }
  return g_src_dir;
}

// ---- code objects on disk ---------------------------------------------------------------------------------------------------------
// A variant takes hipRTC 0.4 - 20 s (the workgroup kernels with a drift compiled in: 10 - 20 s) and was paid again by every process.  The
// code object now also goes to a cache directory -- CDKF_RTC_CACHE_DIR, else rtc_cache/ beside the library (in-tree: it travels with the
// built library), else ~/.cache/cdkf_rtc -- under a key that covers everything the output depends on: the generated source, the
// compile options (target, optimisation level), the hipRTC version, and the CONTENT of every header the source can include (all *.h /
// *.inc of the kernel directory: a changed kernel invalidates every entry).  Files are written to a temporary name and renamed, so
// concurrent processes (ranks, pytest workers) never see a partial entry; a damaged or foreign file fails its header check and is
// recompiled over.  CDKF_RTC_CACHE=0 turns the cache off.
struct Fnv128 {  // two independent 64-bit FNV-1a streams (different offsets / an extra rotation): 128 bits of key
  uint64_t a = 1469598103934665603ull, b = 0x9e3779b97f4a7c15ull;
  void feed(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
      a = (a ^ c[i]) * 1099511628211ull;
      b = ((b << 5) | (b >> 59)) ^ (c[i] + 0x100 * (i & 0xff));
      b *= 0x100000001b3ull;
    }
  }
  void feed(const std::string& s) {
    const uint64_t n = s.size();
    feed(&n, sizeof(n));
    feed(s.data(), s.size());
  }
};

bool rtc_cache_enabled() {
  const char* e = getenv("CDKF_RTC_CACHE");
  return !(e && e[0] == '0');
}

// hash of every header under the kernel directory: computed once per directory (cdkf_set_kernel_source_dir may move it), the
// directory's path part of it; a header that cannot be read yields an EMPTY digest = "do not use the cache" 
std::string rtc_headers_digest() {
  static std::mutex mu;
  static std::map<std::string, std::string> by_dir;
  const std::string dir = source_dir();
  std::lock_guard<std::mutex> lock(mu);
  auto it = by_dir.find(dir);
  if (it != by_dir.end()) return it->second;
  std::string digest;
  {
    Fnv128 h;
    bool unreadable = false;
    h.feed(dir);
    std::vector<std::string> names;
    if (DIR* dp = opendir(dir.c_str())) {
      while (dirent* e = readdir(dp)) {
        const std::string nm = e->d_name;
        const size_t dot = nm.find_last_of('.');
        if (dot == std::string::npos) continue;
        const std::string ext = nm.substr(dot);
        if (ext == ".h" || ext == ".inc") names.push_back(nm);
      }
      closedir(dp);
    }
    std::sort(names.begin(), names.end());
    for (const std::string& nm : names) {
      h.feed(nm);
      if (FILE* f = fopen((dir + "/" + nm).c_str(), "rb")) {
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) h.feed(buf, n);
        fclose(f);
      } else {
        unreadable = true;
      }
    }
    char out[40];
    snprintf(out, sizeof(out), "%016llx%016llx", (unsigned long long)h.a, (unsigned long long)h.b);
    digest = (unreadable || names.empty()) ? std::string() : std::string(out);
  }
  by_dir[dir] = digest;
  return digest;
}

std::string rtc_cache_dir() {
  static std::string dir;
  static std::once_flag once;
  std::call_once(once, [] {
    auto usable = [](const std::string& d) {
      if (d.empty()) return false;
      (void)mkdir(d.c_str(), 0755);
      struct stat sb;
      // (what is stored there is loaded as GPU code: a directory others can write to is not a cache; FNV is a key, not a signature)
      if (stat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || (sb.st_mode & (S_IWGRP | S_IWOTH))) return false;
      return access(d.c_str(), W_OK | X_OK) == 0;
    };
    if (const char* e = getenv("CDKF_RTC_CACHE_DIR")) {
      if (usable(e)) dir = e;
      return;
    }
    const std::string beside = source_dir() + "/../lib/rtc_cache";
    if (usable(beside)) {
      dir = beside;
      return;
    }
    std::string home;
    if (const char* x = getenv("XDG_CACHE_HOME")) home = x;
    else if (const char* hm = getenv("HOME")) home = std::string(hm) + "/.cache";
    if (!home.empty()) {
      (void)mkdir(home.c_str(), 0755);
      if (usable(home + "/cdkf_rtc")) dir = home + "/cdkf_rtc";
    }
  });
  return dir;
}

// ---- how the run-time compiled kernels are optimised (round 5: what is behind rounds 3 / 4's "-O1 fences"; NOTES.md R5.1) -----------------
// ROCm 7.2's compiler (AMD clang 22.0.0git, roc-7.2.0 26014) miscompiles spill-heavy DOUBLE-PRECISION kernels at -O2 / -O3: a gradient
// that is never stored (forward-sensitivity sweep, d = 2: all 512 registers, 913 spilled, 2.3 KB of scratch per lane), moments 3 % off
// (unscented workgroup kernel, d = 15), NaN (eight-entries-per-thread workgroup kernels, d = 46: ~1400 spilled).  Established
// (scripts/r5_o3_probe.py, scripts/r5_mir_delta.py, scripts/r5_spill_table.py; profiles/r05_a_o3_investigation.txt):
//   * the same sources are clean under ASan / UBSan / MSan / TSan in the host build (tests/test_hostsim.py) and agree with the oracle
//     there at -O1 and -O3: not an out-of-bounds index, an uninitialised read or a missing barrier of ours;
//   * pass bisection stops at the pre-RA si-shrink-instructions run in every case, but its rewrites only perturb the register allocation
//     that follows: applied to the machine IR in subsets no single instruction is at fault (delta debugging stays wrong down to 2 520 of
//     5 760 changed lines and no chunk alone is wrong); -verify-machineinstrs passes on the wrong build;
//   * every wrong build on record spills >= 700 vector registers to scratch memory; no build with <= 400 has been wrong at any level
//     (the -O3 filters of d <= 6: 0 - 270, every -O1 build: 0 - 396);
//   * single -mllvm switches flip individual kernels (the basic VGPR allocator, sub-register liveness off, deferred spilling, wave-uniform
//     step loops: CDKF_RTC_UNIFORM) but none is right everywhere -- AND NONE CAN BE CHOSEN PER KERNEL: hipRTC parses LLVM's -mllvm options
//     ONCE PER PROCESS, the first compilation's set (or its absence) is frozen for every later one (measured: the spill count of the same
//     source under [-O3 basic, -O3, -O3 basic] = 1621, 1621, 1621 and under [-O3, -O3 basic, -O3] = 891, 891, 891).  A policy that
//     depends on an -mllvm option therefore depends on which kernel a process happens to compile first, and poisons the disk cache.
//   * LOCATED (rocgdb value trace -> machine IR; rtc_exec_prologue_defect below, profiles/r05_j_root_cause.txt): the vector allocator's
//     split / spill code is inserted at the top of a flow (join) block IN FRONT of the exec restore; loop-carried spill slots go stale.
// So the shipped policy uses the optimisation LEVEL only, and decides it on the code object it gets: a register-resident variant is
// built at -O3; if its machine code shows that shape, or its metadata reports more than CDKF_RTC_SPILL_LIMIT (default 300) spilled
// vector registers (the regime every wrong build was in), that build is discarded and the variant rebuilt at -O1 (which spills 2 - 80 x
// less on these kernels), checked again, and refused if the shape is still there.  The workgroup variants stay at -O1 (no wrong result in
// 700 + random problems; -O3 is wrong on the d = 15 unscented case).  CDKF_RTC_POLICY = o1 | o3 forces a level (A/B; "o3" is the canary of
// tests/test_gpu_toolchain.py); o3basic | o1basic | o3subreg add the -mllvm switch named -- meaningful only as the FIRST compilation of
// a process (scripts/r5_o3_probe.py runs every case in a process of its own).
struct RtcPolicy {
  const char* olevel;
  const char* extra1;  // "-mllvm" or null
  const char* extra2;
  long spill_limit;    // > 0: a build that spills more vector registers than this is rebuilt at -O1 (0: keep whatever comes out)
};
RtcPolicy rtc_policy(bool workgroup) {
  const char* e = getenv("CDKF_RTC_POLICY");
  const std::string p = e ? e : "";
  if (p == "o1") return {"-O1", nullptr, nullptr, 0};
  if (p == "o3") return {"-O3", nullptr, nullptr, 0};
  if (p == "o3basic") return {"-O3", "-mllvm", "-vgpr-regalloc=basic", 0};
  if (p == "o1basic") return {"-O1", "-mllvm", "-vgpr-regalloc=basic", 0};
  if (p == "o3subreg") return {"-O3", "-mllvm", "-enable-subreg-liveness=0", 0};
  if (workgroup) return {"-O1", nullptr, nullptr, 0};
  long lim = 300;
  if (const char* l = getenv("CDKF_RTC_SPILL_LIMIT")) lim = atol(l) > 0 ? atol(l) : 300;
  return {"-O3", nullptr, nullptr, lim};
}

// `.vgpr_spill_count` of the (one) kernel of a code object, from its msgpack metadata note: the key is followed by an unsigned integer
// (positive fixint, 0xcc u8, 0xcd u16, 0xce u32 -- big-endian); -1 when the key is not found.
long code_object_vgpr_spills(const std::vector<char>& code) {
  static const char key[] = ".vgpr_spill_count";
  const size_t kl = sizeof(key) - 1;
  long worst = -1;
  for (size_t i = 0; i + kl + 1 < code.size(); ++i) {
    if (memcmp(code.data() + i, key, kl) != 0) continue;
    const unsigned char* q = (const unsigned char*)code.data() + i + kl;
    const size_t left = code.size() - (i + kl);
    long v = -1;
    if (q[0] <= 0x7f) v = q[0];
    else if (q[0] == 0xcc && left > 1) v = q[1];
    else if (q[0] == 0xcd && left > 2) v = ((long)q[1] << 8) | q[2];
    else if (q[0] == 0xce && left > 4) v = ((long)q[1] << 24) | ((long)q[2] << 16) | ((long)q[3] << 8) | q[4];
    if (v > worst) worst = v;
  }
  return worst;
}

// ---- the defect itself, looked for in the machine code (round 5, profiles/r05_j_root_cause.txt) -------------------------------------------
// What the wrong builds have in common, found with rocgdb and a single-step value trace of the d = 2 sweep: the register allocator's
// live-range splitting puts spill stores / VGPR -> AGPR copies at the TOP of a structured-control-flow flow (or join) block -- BEFORE the
// `s_or_saveexec_b64` (`s_or_b64 exec, exec, sX`) that re-enables the lanes of the other arm -- because the scalar allocation that ran
// first left a rematerialised `s_mov_b32` in front of that restore and the block's prologue is then taken to be empty.  The stores run
// under the then-arm's mask (an EMPTY one in the reproducer: every lane takes the other arm of an inlined sin()'s range reduction), the
// loop-carried values they were to save keep stale slots, the Runge-Kutta loop repeats one step until max_steps.  The shape is searched
// for with comgr's disassembler (the library hipRTC itself sits on; symbols taken from the process):
//     s_and_saveexec_b64 sX, c ; [s_xor_b64 sY, exec, sX ;] s_cbranch_execz L ; ... ;
//  L: <lane-masked vector instruction(s)> ; s_or_saveexec_b64 .., sY   |   s_or_b64 exec, exec, sX
// scripts/check_exec_prologue.py is the same rule over llvm-objdump listings (tests/test_exec_prologue.py: the library's objects, the
// cache, the -O3 build of launch_wg8.hip as a second positive).  Returns 1 found, 0 clean, -1 could not look (no comgr / no .text).
struct RtcIns {
  uint64_t addr;
  std::string text;
  uint64_t target;  // branch target, or ~0
};
struct RtcDisCtx {
  const char* base;
  uint64_t size;
  std::string text;
  uint64_t target;
};
static uint64_t rtc_dis_read(uint64_t from, char* to, uint64_t size, void* ud) {
  RtcDisCtx* c = (RtcDisCtx*)ud;
  if (from >= c->size) return 0;
  const uint64_t n = std::min(size, c->size - from);
  std::memcpy(to, c->base + from, n);
  return n;
}
static void rtc_dis_print(const char* ins, void* ud) { ((RtcDisCtx*)ud)->text = ins ? ins : ""; }
static void rtc_dis_addr(uint64_t a, void* ud) { ((RtcDisCtx*)ud)->target = a; }

int rtc_exec_prologue_defect(const std::vector<char>& code, const std::string& arch, std::string* where) {
  typedef struct { uint64_t handle; } info_t;
  typedef int (*create_t)(const char*, uint64_t (*)(uint64_t, char*, uint64_t, void*), void (*)(const char*, void*), void (*)(uint64_t, void*), info_t*);
  typedef int (*dis_t)(info_t, uint64_t, void*, uint64_t*);
  typedef int (*destroy_t)(info_t);
  static create_t create = (create_t)dlsym(RTLD_DEFAULT, "amd_comgr_create_disassembly_info");
  static dis_t dis = (dis_t)dlsym(RTLD_DEFAULT, "amd_comgr_disassemble_instruction");
  static destroy_t destroy = (destroy_t)dlsym(RTLD_DEFAULT, "amd_comgr_destroy_disassembly_info");
  if (!create || !dis || !destroy || code.size() < 64 || std::memcmp(code.data(), "\177ELF", 4) != 0 || code[4] != 2) return -1;
  // the .text section of the ELF64 code object
  auto rd = [&](size_t off, int bytes) { uint64_t v = 0; if (off + bytes <= code.size()) std::memcpy(&v, code.data() + off, bytes); return v; };
  const uint64_t shoff = rd(0x28, 8), shentsize = rd(0x3a, 2), shnum = rd(0x3c, 2), shstrndx = rd(0x3e, 2);
  if (!shoff || shentsize < 64 || shstrndx >= shnum) return -1;
  const uint64_t stroff = rd(shoff + shstrndx * shentsize + 0x18, 8);
  uint64_t toff = 0, tsize = 0;
  for (uint64_t i = 0; i < shnum; ++i) {
    const uint64_t sh = shoff + i * shentsize, name = rd(sh, 4);
    if (stroff + name + 6 <= code.size() && std::memcmp(code.data() + stroff + name, ".text\0", 6) == 0) {
      toff = rd(sh + 0x18, 8);
      tsize = rd(sh + 0x20, 8);
    }
  }
  if (!tsize || toff + tsize > code.size()) return -1;
  RtcDisCtx ctx{code.data() + toff, tsize, std::string(), ~0ull};
  info_t info{0};
  const std::string isa = "amdgcn-amd-amdhsa--" + arch.substr(0, arch.find(':'));
  if (create(isa.c_str(), rtc_dis_read, rtc_dis_print, rtc_dis_addr, &info) != 0) return -1;
  std::vector<RtcIns> ins;
  ins.reserve(tsize / 6);
  for (uint64_t a = 0; a < tsize;) {
    uint64_t n = 0;
    ctx.text.clear();
    ctx.target = ~0ull;
    if (dis(info, a, &ctx, &n) != 0 || n == 0) n = 4;  // (padding / data: step over a dword)
    size_t b = ctx.text.find_first_not_of(" \t");
    ins.push_back({a, b == std::string::npos ? std::string() : ctx.text.substr(b), ctx.target});
    a += n;
  }
  destroy(info);
  auto op_of = [](const std::string& t) { return t.substr(0, t.find_first_of(" \t")); };
  auto toks_of = [](const std::string& t) {
    std::vector<std::string> v;
    std::string cur;
    for (char ch : t) {
      if (ch == ' ' || ch == '\t' || ch == ',') {
        if (!cur.empty()) v.push_back(cur);
        cur.clear();
      } else {
        cur.push_back(ch);
      }
    }
    if (!cur.empty()) v.push_back(cur);
    return v;
  };
  auto starts = [](const std::string& t, const char* p) { return t.compare(0, std::strlen(p), p) == 0; };
  auto masked = [&](const std::string& op) {
    if (starts(op, "scratch_store") || starts(op, "global_store") || starts(op, "buffer_store") || starts(op, "flat_store") || starts(op, "ds_write") ||
        starts(op, "ds_store") || starts(op, "v_accvgpr_write"))
      return true;
    return starts(op, "v_") && !starts(op, "v_readlane") && !starts(op, "v_writelane") && !starts(op, "v_readfirstlane") && !starts(op, "v_cmp");
  };
  std::map<uint64_t, size_t> index;
  for (size_t k = 0; k < ins.size(); ++k) index[ins[k].addr] = k;
  for (size_t k = 0; k < ins.size(); ++k) {
    if (op_of(ins[k].text) != "s_cbranch_execz" || ins[k].target == ~0ull || ins[k].target <= ins[k].addr || !index.count(ins[k].target)) continue;
    std::string sx, sy;
    for (size_t q = k >= 3 ? k - 3 : 0; q < k; ++q) {
      const std::vector<std::string> t = toks_of(ins[q].text);
      if (t.size() >= 4 && t[0] == "s_xor_b64" && t[2] == "exec") sy = t[1];
      if (t.size() >= 2 && t[0] == "s_and_saveexec_b64") sx = t[1];
    }
    const std::string key = !sy.empty() ? sy : sx;
    if (key.empty()) continue;
    int bad = 0;
    for (size_t j = index[ins[k].target]; j < ins.size() && j < index[ins[k].target] + 400; ++j) {
      const std::vector<std::string> t = toks_of(ins[j].text);
      if (t.empty()) break;
      if ((t[0] == "s_or_saveexec_b64" && t.size() >= 3 && t[2] == key) ||
          (t[0] == "s_or_b64" && t.size() >= 4 && t[1] == "exec" && t[2] == "exec" && t[3] == key)) {
        if (bad) {
          if (where) {
            char buf[160];
            snprintf(buf, sizeof(buf), "%d lane-masked instruction(s) in front of `%s` at .text+%#llx", bad, ins[j].text.c_str(), (unsigned long long)ins[j].addr);
            *where = buf;
          }
          return 1;
        }
        break;
      }
      if (starts(t[0], "s_branch") || starts(t[0], "s_cbranch") || starts(t[0], "s_setpc") || starts(t[0], "s_endpgm")) break;
      if (t.size() >= 2 && t[0][0] == 's' && t[1] == key) break;  // (the saved mask is redefined: not the plain shape)
      if (masked(t[0])) ++bad;
    }
  }
  return 0;
}

// wave-uniform step loops in the register-resident variants (cdkf_math.h: CDKF_UNIFORM_INTEGRATE); CDKF_RTC_UNIFORM = 0 | 1 overrides
bool rtc_uniform_steps() {
  const char* e = getenv("CDKF_RTC_UNIFORM");
  if (e && *e) return e[0] == '1';
  return false;
}

// Compiler options every run-time compilation gets beside its optimisation level (part of the cache key).  CDKF_RTC_EXTRA_OPTS
// (space-separated; a debugging aid) adds to them -- to every variant, or with CDKF_RTC_EXTRA_OPTS_ONLY=<text> to the variants whose tag
// contains <text> (tag: the kernel's name expression for the workgroup variants, "reg ukf=<0|1> algo=<0 filter|1 smoother|2 gradient>"
// for the register-resident ones), so that a pass bisection of ONE kernel leaves the other kernels of the same run as shipped.
std::vector<std::string> rtc_extra_options(const std::string& tag) {
  std::vector<std::string> v;
  if (const char* only = getenv("CDKF_RTC_EXTRA_OPTS_ONLY"))
    if (*only && tag.find(only) == std::string::npos) return v;
  if (const char* e = getenv("CDKF_RTC_EXTRA_OPTS")) {
    std::string cur;
    for (const char* p = e;; ++p) {
      if (*p == ' ' || *p == '\0') {
        if (!cur.empty()) v.push_back(cur);
        cur.clear();
        if (!*p) break;
      } else {
        cur.push_back(*p);
      }
    }
  }
  return v;
}

// Compiler investigations (scripts/r5_mir_delta.py): CDKF_RTC_OVERRIDE_CO=<file> is loaded IN PLACE of compiling the variant whose tag
// contains CDKF_RTC_EXTRA_OPTS_ONLY (which must be set) -- a code object built outside the library from the same generated source,
// e.g. by llc from hand-edited machine IR.  The kernel's name and argument list are the generated source's.
bool rtc_override_code(const std::string& tag, std::vector<char>& code) {
  const char* path = getenv("CDKF_RTC_OVERRIDE_CO");
  const char* only = getenv("CDKF_RTC_EXTRA_OPTS_ONLY");
  if (!path || !*path || !only || !*only || tag.find(only) == std::string::npos) return false;
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  std::vector<char> buf;
  char chunk[65536];
  size_t n;
  while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
  fclose(f);
  if (buf.empty()) return false;
  code.swap(buf);
  return true;
}

std::string rtc_cache_key(const std::string& src, const std::string& arch, const char* olevel, const std::string& expr, const std::string& tag) {
  Fnv128 h;
  int maj = 0, min = 0, rtv = 0;
  (void)hiprtcVersion(&maj, &min);
  (void)hipRuntimeGetVersion(&rtv);  // (major / minor / PATCH of the runtime the compiler ships with: a patch release may change code generation)
  const std::string meta = "cdkf-rtc-2|" + arch + "|" + olevel + "|hiprtc " + std::to_string(maj) + "." + std::to_string(min) + " rt " +
                           std::to_string(rtv) + " build " + std::to_string(HIP_VERSION_PATCH) + "|" + rtc_headers_digest() + "|" + expr;
  h.feed(meta);
  for (const std::string& x : rtc_extra_options(tag)) h.feed("|" + x);
  h.feed(src);
  char out[40];
  snprintf(out, sizeof(out), "%016llx%016llx", (unsigned long long)h.a, (unsigned long long)h.b);
  return out;
}

constexpr uint32_t kRtcMagic = 0x43524b43;  // "CKRC"
// how many variants this process took from the disk cache / had to compile (cdkf_rtc_cache_stats: the GPU suite checks the ratio, so
// that a toolchain bump that invalidates the in-tree cache shows up as a message, not as a 2.4x longer run)
std::atomic<long> g_rtc_hits{0}, g_rtc_misses{0};
bool rtc_cache_load(const std::string& key, std::vector<char>& code, std::string& lowered) {
  const std::string dir = rtc_cache_dir();
  if (dir.empty() || !rtc_cache_enabled() || rtc_headers_digest().empty()) return false;
  FILE* f = fopen((dir + "/" + key + ".co").c_str(), "rb");
  if (!f) {
    ++g_rtc_misses;
    return false;
  }
  uint32_t head[4] = {0, 0, 0, 0};  // magic, name bytes, code bytes (low, high)
  bool ok = fread(head, sizeof(head), 1, f) == 1 && head[0] == kRtcMagic && head[1] < 4096;
  const uint64_t nbytes = ok ? ((uint64_t)head[3] << 32) | head[2] : 0;
  ok = ok && nbytes > 0 && nbytes < (1ull << 30);
  if (ok) {
    lowered.assign(head[1], '\0');
    code.resize(nbytes);
    ok = (head[1] == 0 || fread(&lowered[0], head[1], 1, f) == 1) && fread(code.data(), nbytes, 1, f) == 1 && fgetc(f) == EOF;
  }
  fclose(f);
  if (!ok) code.clear();
  ++(ok ? g_rtc_hits : g_rtc_misses);
  return ok;
}
void rtc_cache_store(const std::string& key, const std::vector<char>& code, const std::string& lowered, const std::string& what = std::string()) {
  const std::string dir = rtc_cache_dir();
  if (dir.empty() || !rtc_cache_enabled() || code.empty() || rtc_headers_digest().empty()) return;
  const std::string final_name = dir + "/" + key + ".co", tmp = final_name + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const uint32_t head[4] = {kRtcMagic, (uint32_t)lowered.size(), (uint32_t)(code.size() & 0xffffffffu), (uint32_t)((uint64_t)code.size() >> 32)};
  const bool ok = fwrite(head, sizeof(head), 1, f) == 1 && (lowered.empty() || fwrite(lowered.data(), lowered.size(), 1, f) == 1) &&
                  fwrite(code.data(), code.size(), 1, f) == 1;
  if (fclose(f) != 0 || !ok || rename(tmp.c_str(), final_name.c_str()) != 0) {
    (void)unlink(tmp.c_str());
    return;
  }
  // MANIFEST: one line per stored code object -- key, toolchain, what it is (a text file that can be tracked where the objects are not)
  if (FILE* m = fopen((dir + "/MANIFEST").c_str(), "a")) {
    int maj = 0, min = 0, rtv = 0;
    (void)hiprtcVersion(&maj, &min);
    (void)hipRuntimeGetVersion(&rtv);
    fprintf(m, "%s hiprtc %d.%d runtime %d %zu bytes %s\n", key.c_str(), maj, min, rtv, code.size(), what.c_str());
    fclose(m);
  }
}

// does the snippet name the identifier `w` (a whole word: `t` in "theta" or "tanh" does not count)?
bool has_word(const std::string& src, const std::string& w) {
  auto idc = [](char ch) { return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_'; };
  for (size_t p = src.find(w); p != std::string::npos; p = src.find(w, p + 1)) {
    const bool l = p == 0 || !idc(src[p - 1]), r = p + w.size() >= src.size() || !idc(src[p + w.size()]);
    if (l && r) return true;
  }
  return false;
}
// the drift reads the time (f(x, u, t): stage times reach it through set_time) / the emission does
bool drift_uses_time(const CustomDrift& c) { return has_word(c.f_src, "t") || has_word(c.jac_src, "t") || has_word(c.g_src, "t"); }
bool drift_reads_context(const CustomDrift& c) {
  return drift_uses_time(c) || has_word(c.f_src, "u") || has_word(c.jac_src, "u") || has_word(c.g_src, "u");
}

std::string generate_source(const CustomDrift& c, int bytes, int m, int ukf, int zeroth, int forecast, int smoother,
                            int generic, const CustomEmission* em, int du) {
  std::string s;
  const bool grad = smoother == 2;
  const std::string D_ = std::to_string(c.d), NT_ = std::to_string(c.n_theta > 0 ? c.n_theta : 1);
  const std::string DU_ = std::to_string(du), DU1_ = std::to_string(du > 0 ? du : 1);
  const bool time_dep = drift_uses_time(c);
  // what every snippet sees beside x and theta: the inputs row of the interval and the (stage) time -- f(x, u, t), inference_ekf.py:95, 101-114
  const std::string ctx = "    const R* u = u_; const R t = t_; (void)u; (void)t;\n";
  s += "#define CDKF_UNIFORM_INTEGRATE " + std::string(rtc_uniform_steps() ? "1" : "0") + "\n";
  s += grad ? "#include \"cdkf_grad_kernels.h\"\n" : "#include \"cdkf_reg_kernels.h\"\n";
  s += "#include \"cdkf_dual.h\"\n";
  s += "namespace cdkf {\n";
  s += "template <typename R, int D>\nstruct DriftCustom {\n";
  s += "  static constexpr int NTHETA = " + std::to_string(c.n_theta) + ";\n";
  s += "  static constexpr bool HAS_G = " + std::string(c.has_g ? "true" : "false") + ";\n";
  s += "  static constexpr bool CONST_JAC = false;\n";
  s += "  static constexpr int DU = " + DU_ + ";\n  static constexpr bool TIME = " + std::string(time_dep ? "true" : "false") + ";\n";
  s += "  R th[" + std::to_string(c.n_theta > 0 ? c.n_theta : 1) + "];\n";
  // inputs of the current interval and the time of the current evaluation: written through const references (the argument block travels
  // as one), read by the snippets as `u` and `t`; the dual-number evaluations leave both constant
  s += "  mutable R u_[" + DU1_ + "];\n  mutable R t_;\n";
  s += "  CDKF_DEV void set_time(R tt) const { t_ = tt; }\n";
  s += "  template <typename A_> CDKF_DEV void load_inputs(const A_& a, long n, long k) const {\n"
       "    for (int i_ = 0; i_ < DU; ++i_) u_[i_] = a.u ? a.u[n * a.u_sn + k * a.u_sk + i_ * a.u_si] : R(0);\n  }\n";
  s += "  static constexpr bool nz(int, int) { return true; }\n";
  s += "  CDKF_DEV void f(const R* x, R (&fx)[D]) const {\n    const R* theta = th; (void)theta;\n" + ctx;
  s += "#line 1 \"drift_f\"\n" + c.f_src + "\n  }\n";
  // the same statements with the scalar type T in place of the compute type: T = a dual number differentiates them (cdkf_dual.h)
  if (c.auto_jac || c.auto_g || grad) {
    s += "  template <typename T> CDKF_DEV void f_t(const T* x, const T* theta, T (&fx)[D]) const {\n    (void)theta;\n" + ctx;
    s += "#line 1 \"drift_f\"\n" + c.f_src + "\n  }\n";
  }
  s += "  CDKF_DEV void jac(const R* x, R (&F)[D][D]) const {\n    const R* theta = th; (void)theta;\n" + ctx;
  s += "    for (int i_ = 0; i_ < D; ++i_) for (int j_ = 0; j_ < D; ++j_) F[i_][j_] = R(0);\n";
  if (c.auto_jac) {  // jacfwd(f): D unit directions (inference_ekf.py:95)
    s += "    typedef Dual<R, D> T;\n    T xt[D], tht[" + NT_ + "], ft[D];\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) { xt[i_] = T(x[i_]); xt[i_].g[i_] = R(1); }\n";
    s += "    for (int k_ = 0; k_ < NTHETA; ++k_) tht[k_] = T(th[k_]);\n";
    s += "    f_t<T>(xt, tht, ft);\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) for (int j_ = 0; j_ < D; ++j_) F[i_][j_] = ft[i_].g[j_];\n  }\n";
  } else {
    s += "#line 1 \"drift_jacobian\"\n" + c.jac_src + "\n  }\n";
  }
  s += "  CDKF_DEV void divgrad(const R* x, R (&g)[D]) const {\n    const R* theta = th; (void)theta;\n" + ctx;
  s += "    for (int i_ = 0; i_ < D; ++i_) g[i_] = R(0);\n";
  if (c.auto_g) {  // g_i = d/dx_i sum_j d f_j / d x_j: second derivatives from nested dual numbers (inference_ekf.py:108-116)
    s += "    typedef Dual<R, D> S1;\n    typedef Dual<S1, D> T;\n    T xt[D], tht[" + NT_ + "], ft[D];\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) { xt[i_] = T(x[i_]); xt[i_].v.g[i_] = R(1); xt[i_].g[i_] = S1(R(1)); }\n";
    s += "    for (int k_ = 0; k_ < NTHETA; ++k_) tht[k_] = T(th[k_]);\n";
    s += "    f_t<T>(xt, tht, ft);\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) for (int j_ = 0; j_ < D; ++j_) g[i_] += ft[j_].g[j_].g[i_];\n  }\n";
  } else {
    s += "#line 1 \"drift_divgrad\"\n" + c.g_src + "\n  }\n";
  }
  s += "};\n";
  if (grad) {
    // what the forward-sensitivity sweep needs of the drift for its parameter p (cdkf_grad_kernels.h): directional derivatives of f and
    // of its Jacobian along (dx, e_p dth) -- outer dual: that one direction, inner dual: the D unit directions of the Jacobian
    s += "template <typename R, int D>\nstruct DriftGrad<R, D, DriftCustom<R, D>> {\n";
    s += "  static constexpr int NPAR = DriftCustom<R, D>::NTHETA;\n  static constexpr bool kCurved = false;\n";
    s += "  static CDKF_DEV void curvature(const R*, R*) {}\n  int p;\n  const DriftCustom<R, D>* dr;\n";
    s += "  CDKF_DEV void init(int p_) { p = p_; }\n  CDKF_DEV void bind(const DriftCustom<R, D>& d_) { dr = &d_; }\n";
    s += "  CDKF_DEV void along(const R* x, const R* dx, R dth, R (&fd)[D], R (&Fd)[D][D], bool add) const {\n";
    s += "    typedef Dual<R, 1> S1;\n    typedef Dual<S1, D> T;\n    T xt[D], tht[" + NT_ + "], ft[D];\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) { S1 b(x[i_]); b.g[0] = dx ? dx[i_] : R(0); xt[i_].v = b; "
         "for (int j_ = 0; j_ < D; ++j_) xt[i_].g[j_] = S1(i_ == j_ ? R(1) : R(0)); }\n";
    s += "    for (int k_ = 0; k_ < NPAR; ++k_) { S1 b(dr->th[k_]); b.g[0] = (k_ == p) ? dth : R(0); tht[k_].v = b; "
         "for (int j_ = 0; j_ < D; ++j_) tht[k_].g[j_] = S1(R(0)); }\n";
    s += "    dr->template f_t<T>(xt, tht, ft);\n";
    s += "    for (int i_ = 0; i_ < D; ++i_) { fd[i_] = ft[i_].v.g[0]; for (int j_ = 0; j_ < D; ++j_) "
         "Fd[i_][j_] = (add ? Fd[i_][j_] : R(0)) + ft[i_].g[j_].g[0]; }\n  }\n";
    s += "  CDKF_DEV void dtheta(const R* x, R (&df)[D], R (&dF)[D][D]) const { along(x, nullptr, R(1), df, dF, false); }\n";
    s += "  CDKF_DEV void dstate(const R* x, const R* dm, R (&dF)[D][D]) const { R fd[D]; along(x, dm, R(0), fd, dF, true); }\n";
    s += "};\n";
  }
  s += "}  // namespace cdkf\n";
  if (em) {
    // emission parameters eta = [the model's H block (m x d, row-major) | h_bias (m)], read from the argument block
    s += "namespace cdkf {\ntemplate <typename R, int D, int M>\nstruct EmisCustom {\n";
    s += "  static constexpr bool kCustom = true;\n  R eta_[M * D + M];\n  R u_[" + DU1_ + "];\n  R t_;\n";
    // h(x, u, t0), H(x, u, t0): this step's inputs row and observation time (inference_ekf.py:277-286)
    s += "  template <typename A_> CDKF_DEV void set_ctx(const A_& a, long n, long k, R tt) {\n    t_ = tt;\n"
         "    for (int i_ = 0; i_ < " + DU_ + "; ++i_) u_[i_] = a.u ? a.u[n * a.u_sn + k * a.u_sk + i_ * a.u_si] : R(0);\n"
         "    if (" + DU_ + " == 0) u_[0] = R(0);\n  }\n";
    s += "  template <typename Args> CDKF_DEV void load(const Args& a) {\n";
    s += "    for (int r = 0; r < M; ++r) { for (int k = 0; k < D; ++k) eta_[r * D + k] = a.H[r][k]; eta_[M * D + r] = a.hb[r]; }\n  }\n";
    s += "  CDKF_DEV void h(const R* x, R (&hx)[M]) const {\n    const R* eta = eta_; (void)eta;\n" + ctx;
    s += "#line 1 \"emission_h\"\n" + em->h_src + "\n  }\n";
    const bool auto_hjac = blank(em->jac_src);  // jacfwd(h) by dual numbers (inference_ekf.py:258-259)
    if (auto_hjac) {
      s += "  template <typename T> CDKF_DEV void h_t(const T* x, const T* eta, T (&hx)[M]) const {\n    (void)eta;\n" + ctx;
      s += "#line 1 \"emission_h\"\n" + em->h_src + "\n  }\n";
    }
    s += "  CDKF_DEV void jac(const R* x, R (&H)[M][D]) const {\n    const R* eta = eta_; (void)eta;\n" + ctx;
    s += "    for (int r_ = 0; r_ < M; ++r_) for (int k_ = 0; k_ < D; ++k_) H[r_][k_] = R(0);\n";
    if (auto_hjac) {
      s += "    typedef Dual<R, D> T;\n    T xt[D], et[M * D + M], ht[M];\n";
      s += "    for (int i_ = 0; i_ < D; ++i_) { xt[i_] = T(x[i_]); xt[i_].g[i_] = R(1); }\n";
      s += "    for (int k_ = 0; k_ < M * D + M; ++k_) et[k_] = T(eta_[k_]);\n";
      s += "    h_t<T>(xt, et, ht);\n";
      s += "    for (int r_ = 0; r_ < M; ++r_) for (int k_ = 0; k_ < D; ++k_) H[r_][k_] = ht[r_].g[k_];\n  }\n";
    } else {
      s += "#line 1 \"emission_jacobian\"\n" + em->jac_src + "\n  }\n";
    }
    s += "};\n}  // namespace cdkf\n";
  }
  s += "using R = " + std::string(bytes == 8 ? "double" : "float") + ";\n";
  s += "constexpr int DD = " + std::to_string(c.d) + ", MM = " + std::to_string(m) + ";\n";
  s += "using Drift = cdkf::DriftCustom<R, DD>;\nusing Args = cdkf::RegArgs<R, DD, MM, Drift>;\n";
  // parameter blob (reals): theta | LQL | LQLz | H | hb | Rm | m0 | P0 | dt0 dt_final ukf_c ukf_wm0 ukf_wc0 ukf_wi | rk.a[30] rk.b[6] rk.berr[7] rtol atol c1 c2 c3 | dtmin dtmax safety factormin factormax
  // integer blob (longs) : max_steps order num_iter forecast N T t_sn t_sk y_sn y_sk y_si m_sn m_sk m_si P_sn P_sk P_si stages solver
  //                        adaptive fsal lanes xcd_shift u_sn u_sk u_si
  s += R"(
__device__ __forceinline__ void unpack(Args& a, const R* __restrict__ par, const long* __restrict__ ip, const R* t, const R* y,
                                       R* ll, R* fm, R* fP, R* pm, R* pP, int* status, const R* u) {
  constexpr int NP = cdkf::Dims<DD>::NP;
  int o = 0;
  for (int k = 0; k < Drift::NTHETA; ++k) a.drift.th[k] = par[o + k];
  o += Drift::NTHETA;
  for (int k = 0; k < NP; ++k) a.LQL[k] = par[o + k];
  o += NP;
  for (int k = 0; k < NP; ++k) a.LQLz[k] = par[o + k];
  o += NP;
  for (int r = 0; r < MM; ++r) for (int k = 0; k < DD; ++k) a.H[r][k] = par[o + r * DD + k];
  o += MM * DD;
  for (int r = 0; r < MM; ++r) a.hb[r] = par[o + r];
  o += MM;
  for (int r = 0; r < MM; ++r) for (int c = 0; c < MM; ++c) a.Rm[r][c] = par[o + r * MM + c];
  o += MM * MM;
  for (int k = 0; k < DD; ++k) a.m0[k] = par[o + k];
  o += DD;
  for (int k = 0; k < NP; ++k) a.P0[k] = par[o + k];
  o += NP;
  a.dt0 = par[o]; a.dt_final = par[o + 1]; a.ukf_c = par[o + 2]; a.ukf_wm0 = par[o + 3]; a.ukf_wc0 = par[o + 4];
  a.ukf_wi = par[o + 5];
  o += 6;
  for (int s = 0; s < 6; ++s) for (int j = 0; j < 5; ++j) a.rk.a[s][j] = par[o + s * 5 + j];
  o += 30;
  for (int s = 0; s < 6; ++s) a.rk.b[s] = par[o + s];
  o += 6;
  for (int s = 0; s < 7; ++s) a.rk.berr[s] = par[o + s];
  o += 7;
  a.rk.rtol = par[o]; a.rk.atol = par[o + 1]; a.rk.c1 = par[o + 2]; a.rk.c2 = par[o + 3]; a.rk.c3 = par[o + 4];
  a.rk.dtmin = par[o + 5]; a.rk.dtmax = par[o + 6]; a.rk.safety = par[o + 7]; a.rk.fmin = par[o + 8]; a.rk.fmax = par[o + 9];
  a.rk.stages = (int)ip[17]; a.solver = (int)ip[18]; a.rk.adaptive = (int)ip[19]; a.rk.fsal = (int)ip[20]; a.lanes = (int)ip[21]; a.xcd_shift = (int)ip[22];
  a.max_steps = ip[0]; a.order = (int)ip[1]; a.num_iter = (int)ip[2]; a.forecast = (int)ip[3]; a.N = ip[4]; a.T = ip[5];
  a.t_sn = ip[6]; a.t_sk = ip[7]; a.y_sn = ip[8]; a.y_sk = ip[9]; a.y_si = ip[10]; a.m_sn = ip[11]; a.m_sk = ip[12];
  a.m_si = ip[13]; a.P_sn = ip[14]; a.P_sk = ip[15]; a.P_si = ip[16];
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.pm = pm; a.pP = pP; a.status = status;
  a.u = u; a.u_sn = ip[23]; a.u_sk = ip[24]; a.u_si = ip[25];
  a.drift.t_ = R(0);
  for (int k = 0; k < (Drift::DU > 0 ? Drift::DU : 1); ++k) a.drift.u_[k] = R(0);
}
)";
  // CDKF_RTC_KERNEL_ATTR: extra attributes on the generated register-resident kernel (compiler investigations, e.g.
  // "__attribute__((amdgpu_waves_per_eu(2,2)))" to halve its register budget); part of the source, hence of the cache key
  const std::string kattr = getenv("CDKF_RTC_KERNEL_ATTR") ? std::string(getenv("CDKF_RTC_KERNEL_ATTR")) + " " : std::string();
  if (grad) {
    s += "extern \"C\" __global__ __launch_bounds__(64, 1) " + kattr + "void cdkf_custom_kernel(const R* par, const long* ip, const R* t, "
         "const R* y, R* ll, R* fm, R* fP, R* pm, R* pP, int* status, R* sm, R* sP, const R* u) {\n  cdkf::GradArgs<R, DD, MM, Drift> ga;\n"
         "  unpack(ga.a, par, ip, t, y, ll, nullptr, nullptr, nullptr, nullptr, status, u);\n  ga.grad = fm;\n";
    s += "  cdkf::ekf_grad_reg_body<R, DD, MM, Drift, " + std::string(generic ? "true" : "false") + ", false>(ga);\n}\n";
  } else if (!smoother) {
    s += "extern \"C\" __global__ __launch_bounds__(64, 1) " + kattr + "void cdkf_custom_kernel(const R* par, const long* ip, const R* t, "
         "const R* y, R* ll, R* fm, R* fP, R* pm, R* pP, int* status, R* sm, R* sP, const R* u) {\n  Args a;\n"
         "  unpack(a, par, ip, t, y, ll, fm, fP, pm, pP, status, u);\n";
    s += "  cdkf::filter_reg_body<R, DD, MM, Drift, " + std::string(ukf ? "true" : "false") + ", " +
         std::string(zeroth ? "true" : "false") + ", false, cdkf::kOutSome, " + std::string(forecast ? "true" : "false") + ", " +
         std::string(generic ? "true" : "false") + (em ? ", cdkf::EmisCustom<R, DD, MM>" : "") + ">(a);\n}\n";
  } else {
    s += "extern \"C\" __global__ __launch_bounds__(64) void cdkf_custom_kernel(const R* par, const long* ip, const R* t, "
         "const R* y, R* ll, R* fm, R* fP, R* pm, R* pP, int* status, R* sm, R* sP, const R* u) {\n  Args a;\n"
         "  unpack(a, par, ip, t, y, ll, fm, fP, pm, pP, status, u);\n"
         "  cdkf::ekf_smoother_reg_body<R, DD, MM, Drift, " + std::string(generic ? "true" : "false") + ">(a, sm, sP);\n}\n";
  }
  return s;
}

// compile one variant; `code` receives the code object.  No GPU needed (the target is named explicitly).
int compile_variant(const CustomDrift& c, const Key& key, const std::string& arch, std::vector<char>& code) {
  CustomEmission em;
  const int ek = std::get<8>(key);
  if (ek) {
    std::lock_guard<std::mutex> lock(g_mutex_emis());
    em = g_emis[ek - CDKF_EMISSION_CUSTOM_BASE];
  }
  const std::string src = generate_source(c, std::get<1>(key), std::get<2>(key), std::get<3>(key), std::get<4>(key),
                                          std::get<5>(key), std::get<6>(key), std::get<7>(key), ek ? &em : nullptr, std::get<9>(key));
  // (rounds 3 / 4 built the forward-sensitivity sweep at -O1 after wrong gradients at -O2 / -O3: see rtc_policy above for the rule now)
  const RtcPolicy pol = rtc_policy(false);
  const char* olevel = pol.olevel;
  const std::string tag = "reg ukf=" + std::to_string(std::get<3>(key)) + " algo=" + std::to_string(std::get<6>(key));
  if (rtc_override_code(tag, code)) return CDKF_OK;
  const std::string pol_text = std::string(olevel) + (pol.extra2 ? pol.extra2 : "") +
                               (pol.spill_limit ? "/spills>" + std::to_string(pol.spill_limit) + "->-O1" : std::string());
  const std::string cache_key = rtc_cache_key(src, arch, pol_text.c_str(), "cdkf_custom_kernel", tag);
  {
    std::string unused;
    if (!getenv("CDKF_CUSTOM_DUMP") && rtc_cache_load(cache_key, code, unused)) return CDKF_OK;
  }
  long spills = -1;
  std::string defect_note;
  for (int attempt = 0; attempt < 2; ++attempt) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "cdkf_custom_drift.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      set_error("custom drift: hiprtcCreateProgram failed");
      return CDKF_EHIP;
    }
    const std::string inc = "-I" + source_dir(), off = "--offload-arch=" + arch;
    const std::vector<std::string> extra = rtc_extra_options(tag);
    std::vector<const char*> opts = {off.c_str(), olevel, "-std=c++17", inc.c_str(), "-Wno-pass-failed"};
    if (pol.extra1) {
      opts.push_back(pol.extra1);
      opts.push_back(pol.extra2);
    }
    for (const std::string& x : extra) opts.push_back(x.c_str());
    const hiprtcResult res = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (res != HIPRTC_SUCCESS) {
      size_t n = 0;
      hiprtcGetProgramLogSize(prog, &n);
      std::string log(n ? n : 1, '\0');
      if (n) hiprtcGetProgramLog(prog, &log[0]);
      // keep the diagnostics, drop the include-chain preamble
      std::string brief;
      size_t pos = 0;
      while (pos < log.size()) {
        size_t eol = log.find('\n', pos);
        if (eol == std::string::npos) eol = log.size();
        if (log.compare(pos, 21, "In file included from") != 0) brief.append(log, pos, eol - pos + 1);
        pos = eol + 1;
      }
      set_error("custom drift: compilation failed (%s): %.400s", hiprtcGetErrorString(res), brief.c_str());
      hiprtcDestroyProgram(&prog);
      return CDKF_EINVAL;
    }
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    code.resize(sz);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    const long now = code_object_vgpr_spills(code);
    if (attempt == 0) spills = now;
    std::string where;
    const int defect = getenv("CDKF_RTC_NO_DEFECT_CHECK") ? 0 : rtc_exec_prologue_defect(code, arch, &where);
    // the rule of rtc_policy: a build that shows the defect's shape, or is past the spill limit (the regime every wrong build was in),
    // is not trusted at this level -- once more at -O1 (unknown spill count: the same)
    if (attempt == 0 && pol.spill_limit > 0 && (defect == 1 || now < 0 || now > pol.spill_limit) && std::string(olevel) != "-O1") {
      if (defect == 1) defect_note = " exec-prologue defect at " + std::string(olevel) + ": " + where + ";";
      olevel = "-O1";
      continue;
    }
    if (defect == 1 && pol.spill_limit > 0) {  // (a forced level -- CDKF_RTC_POLICY -- is taken as asked: the canary needs the wrong build)
      set_error("custom drift: the compiler placed vector spill code in front of an execution-mask restore at -O3 AND at -O1 (%s) -- a known "
                "ROCm 7.2 register-allocation defect that yields wrong results; refusing to run this kernel", where.c_str());
      return CDKF_EUNSUPPORTED;
    }
    break;
  }
  rtc_cache_store(cache_key, code, std::string(), tag + " " + olevel + " (vgpr spills at " + pol.olevel + ": " + std::to_string(spills) + ";" + defect_note + ")");
  if (const char* dir = getenv("CDKF_CUSTOM_DUMP")) {  // debugging aid: the generated source and its code object
    const std::string base = std::string(dir) + "/cdkf_custom_reg_" + std::to_string(std::get<1>(key)) + "_m" + std::to_string(std::get<2>(key)) + "_" +
                             std::to_string(std::get<3>(key)) + "_" + std::to_string(std::get<4>(key)) + "_" + std::to_string(std::get<6>(key)) + "_" + cache_key.substr(0, 8);
    if (FILE* f = fopen((base + ".hip").c_str(), "w")) {
      fwrite(src.data(), 1, src.size(), f);
      fclose(f);
    }
    if (FILE* f = fopen((base + ".co").c_str(), "wb")) {
      fwrite(code.data(), 1, code.size(), f);
      fclose(f);
    }
  }
  return CDKF_OK;
}

int get_function(int kind, const Key& key, hipFunction_t* fn) {
  std::lock_guard<std::mutex> lock(g_mutex);
  int dev = 0;
  CDKF_HIP_CHECK(hipGetDevice(&dev));
  auto it = g_modules.find({dev, key});
  if (it != g_modules.end()) {
    *fn = it->second.fn;
    return CDKF_OK;
  }
  const CustomDrift& c = g_drifts[kind - CDKF_DRIFT_CUSTOM_BASE];
  hipDeviceProp_t prop;
  CDKF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  std::vector<char> code;
  int rc = compile_variant(c, key, prop.gcnArchName, code);
  if (rc) return rc;
  Compiled m;
  CDKF_HIP_CHECK(hipModuleLoadData(&m.module, code.data()));
  CDKF_HIP_CHECK(hipModuleGetFunction(&m.fn, m.module, "cdkf_custom_kernel"));
  g_modules[{dev, key}] = m;
  *fn = m.fn;
  return CDKF_OK;
}

// ---- beyond six dimensions: the workgroup-per-trajectory kernels with the drift compiled in -------------------------------------
// kind, bytes per real, entries per thread, ukf, smoother, LDS bytes
using WgKey = std::tuple<int, int, int, int, int, long, int>;  // (..., input_dim)
std::map<std::pair<int, WgKey>, Compiled> g_wg_modules;

std::string generate_wg_source(const CustomDrift& c, size_t lds, int du) {
  std::string s;
  const std::string D_ = std::to_string(c.d), NT_ = std::to_string(c.n_theta > 0 ? c.n_theta : 1), NTH_ = std::to_string(c.n_theta);
  const bool second = c.has_g && c.auto_g;  // (registered as identically zero: has_g with a blank source -- nothing to add)
  s += "#define CDKF_WG_CUSTOM 1\n#define CDKF_WG_CUSTOM_SECOND " + std::string(second ? "1" : "0") + "\n";
  s += "#define CDKF_WG_CUSTOM_TIME " + std::string(drift_uses_time(c) ? "1" : "0") + "\n";
  s += "#define CDKF_WG_STATIC_LDS " + std::to_string(lds) + "\n";
  s += "#include \"cdkf_reg_kernels.h\"\n#include \"cdkf_wg2_kernels.h\"\n#include \"cdkf_dual.h\"\nnamespace cdkf {\n";
  s += "constexpr int CD = " + D_ + ", CNT = " + NT_ + ", CNTH = " + NTH_ + ", CDU = " + std::to_string(du) + ", CDU1 = " + std::to_string(du > 0 ? du : 1) + ";\n";
  // (u: this interval's inputs row, t: the time of the evaluation -- f(x, u, t), inference_ekf.py:95; constants under the dual numbers)
  s += "// (R: the compute type, as in the register-resident kernels' DriftCustom<R, D> -- the snippet may write R(...) constants)\ntemplate <typename R, typename T> __device__ __forceinline__ void custom_f(const T* x, const T* theta, T (&fx)[CD], const R* u, const R t) {\n  (void)theta; (void)u; (void)t;\n";
  s += "#line 1 \"drift_f\"\n" + c.f_src + "\n}\n";
  s += R"(
// the inputs row of the interval in hand (WgArgs::ctx_uoff) into registers; the evaluation's time is a.ctx_t
template <typename R>
__device__ __forceinline__ void wg_custom_ctx(const WgArgs<R>& a, R (&ub)[CDU1]) {
  ub[0] = R(0);
  for (int i = 0; i < CDU; ++i) ub[i] = a.u ? a.u[a.ctx_uoff + i * a.u_si] : R(0);
}

template <typename R>
__device__ void wg_custom_drift(const WgArgs<R>& a, const WgLds<R>& L, const R* x, R* fv, R* F, R* gv) {
  const R* th = a.par + a.o_theta;
  const int lq = a.lq;
  R ub[CDU1];
  wg_custom_ctx(a, ub);
  const R tt = a.ctx_t;
  if (!F) {
    if (threadIdx.x == 0) {
      R xr[CD], thr[CNT], fr[CD];
      for (int i = 0; i < CD; ++i) xr[i] = x[i];
      for (int k = 0; k < CNTH; ++k) thr[k] = th[k];
      custom_f<R, R>(xr, thr, fr, ub, tt);
      for (int i = 0; i < CD; ++i) fv[i] = fr[i];
    }
  } else {  // jacfwd(f) (inference_ekf.py:95): thread j carries the unit direction e_j -> column j of the Jacobian
    CDKF_WG_FOR(j, CD) {
      typedef Dual<R, 1> T;
      T xt[CD], tht[CNT], ft[CD];
      for (int i = 0; i < CD; ++i) {
        xt[i] = T(x[i]);
        xt[i].g[0] = (i == j) ? R(1) : R(0);
      }
      for (int k = 0; k < CNTH; ++k) tht[k] = T(th[k]);
      custom_f<R, T>(xt, tht, ft, ub, tt);
      for (int i = 0; i < CD; ++i) F[i * lq + j] = ft[i].g[0];
      if (j == 0)
        for (int i = 0; i < CD; ++i) fv[i] = ft[i].v;
    }
  }
  if (gv) {
#if CDKF_WG_CUSTOM_SECOND
    // g_k = d/dx_k sum_i d f_i / d x_i (inference_ekf.py:108-116): one nested-dual evaluation per pair (i, k), summed over i afterwards
    R* W = L.mat(L.plan.i_A);  // free until the right-hand side's product F Ps is formed
    CDKF_WG_FOR(e, CD * CD) {
      const int i = e / CD, k = e - i * CD;
      typedef Dual<R, 1> S1;
      typedef Dual<S1, 1> T;
      T xt[CD], tht[CNT], ft[CD];
      for (int l = 0; l < CD; ++l) {
        S1 b(x[l]);
        b.g[0] = (l == k) ? R(1) : R(0);
        xt[l].v = b;
        xt[l].g[0] = S1(l == i ? R(1) : R(0));
      }
      for (int kk = 0; kk < CNTH; ++kk) {
        tht[kk].v = S1(th[kk]);
        tht[kk].g[0] = S1(R(0));
      }
      custom_f<R, T>(xt, tht, ft, ub, tt);
      R v = R(0);
      for (int l = 0; l < CD; ++l)
        if (l == i) v = ft[l].g[0].g[0];
      W[i * lq + k] = v;
    }
    __syncthreads();
    CDKF_WG_FOR(k, CD) {
      R sum = R(0);
      for (int i = 0; i < CD; ++i) sum += W[i * lq + k];
      gv[k] = sum;
    }
#else
    CDKF_WG_FOR(k, CD) gv[k] = R(0);
#endif
  }
}

template <typename R>
__device__ void wg_custom_sigma(const WgArgs<R>& a, const WgLds<R>& L, const R* ms, const R* O, R* f0, R* DF, R* foo) {
  const R* th = a.par + a.o_theta;
  const int lq = a.lq;
  (void)L;
  R ub[CDU1];
  wg_custom_ctx(a, ub);
  const R tt = a.ctx_t;
  CDKF_WG_FOR(i, CD + 1) {
    R xr[CD], thr[CNT], fp[CD], fm[CD];
    for (int k = 0; k < CNTH; ++k) thr[k] = th[k];
    if (i == CD) {
      for (int r = 0; r < CD; ++r) xr[r] = ms[r];
      custom_f<R, R>(xr, thr, fp, ub, tt);
      for (int r = 0; r < CD; ++r) f0[r] = fp[r];
    } else {
      for (int r = 0; r < CD; ++r) xr[r] = ms[r] + O[r * lq + i];
      custom_f<R, R>(xr, thr, fp, ub, tt);
      for (int r = 0; r < CD; ++r) xr[r] = ms[r] - O[r * lq + i];
      custom_f<R, R>(xr, thr, fm, ub, tt);
      for (int r = 0; r < CD; ++r) {
        DF[r * lq + i] = fp[r] - fm[r];
        foo[r * lq + i] = fp[r] + fm[r];
      }
    }
  }
}
}  // namespace cdkf
)";
  return s;
}

// the shape-generic reverse sweep (cdkf_adjoint_wg_kernels.h) with the drift compiled in: d ll / d theta and every other leaf
std::string generate_awg_source(const CustomDrift& c, size_t lds, int du) {
  std::string s;
  const std::string D_ = std::to_string(c.d), NT_ = std::to_string(c.n_theta > 0 ? c.n_theta : 1), NTH_ = std::to_string(c.n_theta);
  s += "#define CDKF_AWG_CUSTOM " + NTH_ + "\n#define CDKF_WG_STATIC_LDS " + std::to_string(lds) + "\n";
  s += "#define CDKF_AWG_CUSTOM_DU " + std::to_string(du) + "\n";
  s += "#define CDKF_AWG_CUSTOM_SECOND " + std::string((c.has_g && c.auto_g) ? "1" : "0") + "\n";
  s += "#include \"cdkf_reg_kernels.h\"\n#include \"cdkf_adjoint_wg_kernels.h\"\n#include \"cdkf_dual.h\"\nnamespace cdkf {\n";
  s += "constexpr int CD = " + D_ + ", CNT = " + NT_ + ", CNTH = " + NTH_ + ";\n";
  s += "// (R: the compute type, as in the register-resident kernels' DriftCustom<R, D> -- the snippet may write R(...) constants;\n// u, t: the interval's inputs row and the evaluation's time, constants under the dual numbers)\ntemplate <typename R, typename T> __device__ __forceinline__ void custom_f(const T* x, const T* theta, T (&fx)[CD], const R* u, const R t) {\n  (void)theta; (void)u; (void)t;\n";
  s += "#line 1 \"drift_f\"\n" + c.f_src + "\n}\n";
  s += R"(
template <typename R>
__device__ void awg_custom_column(const R* th, const R* x, int j, R* F, int ld, R* fv, const R* uin, R tin) {
  typedef Dual<R, 1> T;
  T xt[CD], tht[CNT], ft[CD];
  for (int i = 0; i < CD; ++i) {
    xt[i] = T(x[i]);
    xt[i].g[0] = (i == j) ? R(1) : R(0);
  }
  for (int k = 0; k < CNTH; ++k) tht[k] = T(th[k]);
  custom_f<R, T>(xt, tht, ft, uin, tin);
  for (int i = 0; i < CD; ++i) F[i * ld + j] = ft[i].g[0];
  if (fv)
    for (int i = 0; i < CD; ++i) fv[i] = ft[i].v;
}

// inner dual: the direction e_j of the Jacobian's column; outer dual: the state component or parameter z
template <typename R>
__device__ __forceinline__ R awg_custom_contract(const R* th, const R* x, const R* G, int ld, int j, int z, const R* lam, const R* uin, R tin) {
  typedef Dual<R, 1> S1;
  typedef Dual<S1, 1> T;
  T xt[CD], tht[CNT], ft[CD];
  for (int l = 0; l < CD; ++l) {
    S1 b(x[l]);
    b.g[0] = (l == j) ? R(1) : R(0);
    xt[l].v = b;
    xt[l].g[0] = S1(l == z ? R(1) : R(0));
  }
  for (int k = 0; k < CNTH; ++k) {
    tht[k].v = S1(th[k]);
    tht[k].g[0] = S1(k + CD == z ? R(1) : R(0));
  }
  custom_f<R, T>(xt, tht, ft, uin, tin);
  R s = R(0);
  for (int i = 0; i < CD; ++i) s += G[i * ld + j] * ft[i].g[0].g[0];
  if (lam)
    for (int i = 0; i < CD; ++i) s += lam[i] * ft[i].g[0].v;
  return s;
}

template <typename R>
__device__ __forceinline__ R awg_custom_divpair(const R* th, const R* x, int i, int k, const R* uin, R tin) {
  typedef Dual<R, 1> S1;
  typedef Dual<S1, 1> T;
  T xt[CD], tht[CNT], ft[CD];
  for (int l = 0; l < CD; ++l) {
    S1 b(x[l]);
    b.g[0] = (l == k) ? R(1) : R(0);
    xt[l].v = b;
    xt[l].g[0] = S1(l == i ? R(1) : R(0));
  }
  for (int kk = 0; kk < CNTH; ++kk) {
    tht[kk].v = S1(th[kk]);
    tht[kk].g[0] = S1(R(0));
  }
  custom_f<R, T>(xt, tht, ft, uin, tin);
  R v = R(0);
  for (int l = 0; l < CD; ++l)
    if (l == i) v = ft[l].g[0].g[0];
  return v;
}

// innermost dual: the direction u; middle: e_i; outer: the state component or parameter z
template <typename R>
__device__ __forceinline__ R awg_custom_third(const R* th, const R* x, const R* u, int i, int z, const R* uin, R tin) {
  typedef Dual<R, 1> S1;
  typedef Dual<S1, 1> S2;
  typedef Dual<S2, 1> T;
  T xt[CD], tht[CNT], ft[CD];
  for (int l = 0; l < CD; ++l) {
    S1 a_(x[l]);
    a_.g[0] = u[l];
    S2 b;
    b.v = a_;
    b.g[0] = S1(l == i ? R(1) : R(0));
    xt[l].v = b;
    xt[l].g[0] = S2(l == z ? R(1) : R(0));
  }
  for (int kk = 0; kk < CNTH; ++kk) {
    tht[kk].v = S2(th[kk]);
    tht[kk].g[0] = S2(kk + CD == z ? R(1) : R(0));
  }
  custom_f<R, T>(xt, tht, ft, uin, tin);
  R v = R(0);
  for (int l = 0; l < CD; ++l)
    if (l == i) v = ft[l].g[0].g[0].g[0];
  return v;
}
}  // namespace cdkf
)";
  return s;
}

std::string wg_kernel_expr(int bytes, int ept, int ukf, int smoother) {
  const std::string R_ = bytes == 8 ? "double" : "float";
  if (smoother == 2) return "cdkf::ekf_adjoint_wg_kernel<" + R_ + ", " + std::to_string(ept) + ">";
  if (smoother) return "cdkf::ekf_smoother_wg_kernel<" + R_ + ", " + std::to_string(ept) + ">";
  return "cdkf::ekf_filter_wg_kernel<" + R_ + ", " + std::to_string(ept) + ", " + (ukf ? "true" : "false") + ", cdkf::kDriftAny>";
}

int compile_wg_variant(const CustomDrift& c, const WgKey& key, const std::string& arch, std::vector<char>& code, std::string& lowered) {
  const int bytes = std::get<1>(key), ept = std::get<2>(key), ukf = std::get<3>(key), smoother = std::get<4>(key);
  const std::string src = smoother == 2 ? generate_awg_source(c, (size_t)std::get<5>(key), std::get<6>(key))
                                        : generate_wg_source(c, (size_t)std::get<5>(key), std::get<6>(key));
  const std::string expr = wg_kernel_expr(bytes, ept, ukf, smoother);
  // (round 4 built every workgroup variant at -O1 after the unscented d = 15 kernel came out 3 % off at -O3: see rtc_policy for the cause)
  const RtcPolicy pol = rtc_policy(true);
  const char* olevel = pol.olevel;
  const std::string cache_key = rtc_cache_key(src, arch, (std::string(olevel) + (pol.extra2 ? pol.extra2 : "")).c_str(), expr, expr);
  if (!getenv("CDKF_CUSTOM_DUMP") && rtc_cache_load(cache_key, code, lowered) && !lowered.empty()) return CDKF_OK;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "cdkf_custom_drift_wg.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    set_error("custom drift: hiprtcCreateProgram failed");
    return CDKF_EHIP;
  }
  if (hiprtcAddNameExpression(prog, expr.c_str()) != HIPRTC_SUCCESS) {
    set_error("custom drift: hiprtcAddNameExpression failed");
    hiprtcDestroyProgram(&prog);
    return CDKF_EHIP;
  }
  // (the instantiations with eight or more entries per thread are built at -O1 in the library too: launch_wg8.hip, Makefile)
  const std::string inc = "-I" + source_dir(), off = "--offload-arch=" + arch;
  // -O1: the wg kernels' instantiations with eight or more entries per thread, as in the library (launch_wg8.hip, Makefile)
  const std::vector<std::string> extra = rtc_extra_options(expr);
  std::vector<const char*> opts = {off.c_str(), olevel, "-std=c++17", inc.c_str(), "-Wno-pass-failed"};
  if (pol.extra1) {
    opts.push_back(pol.extra1);
    opts.push_back(pol.extra2);
  }
  for (const std::string& x : extra) opts.push_back(x.c_str());
  const hiprtcResult res = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (res != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n ? n : 1, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    std::string brief;
    size_t pos = 0;
    while (pos < log.size()) {
      size_t eol = log.find('\n', pos);
      if (eol == std::string::npos) eol = log.size();
      if (log.compare(pos, 21, "In file included from") != 0) brief.append(log, pos, eol - pos + 1);
      pos = eol + 1;
    }
    set_error("custom drift: compilation failed (%s): %.400s", hiprtcGetErrorString(res), brief.c_str());
    hiprtcDestroyProgram(&prog);
    return CDKF_EINVAL;
  }
  const char* name = nullptr;
  if (hiprtcGetLoweredName(prog, expr.c_str(), &name) != HIPRTC_SUCCESS || !name) {
    set_error("custom drift: no lowered name for %s", expr.c_str());
    hiprtcDestroyProgram(&prog);
    return CDKF_EHIP;
  }
  lowered = name;
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  code.resize(sz);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  {
    std::string where;
    if (!getenv("CDKF_RTC_NO_DEFECT_CHECK") && !getenv("CDKF_RTC_POLICY") && rtc_exec_prologue_defect(code, arch, &where) == 1) {
      set_error("custom drift (workgroup kernels): the compiler placed vector spill code in front of an execution-mask restore at %s (%s) -- a "
                "known ROCm 7.2 register-allocation defect that yields wrong results; refusing to run this kernel", olevel, where.c_str());
      return CDKF_EUNSUPPORTED;
    }
  }
  rtc_cache_store(cache_key, code, lowered, expr + " " + olevel);
  if (const char* dir = getenv("CDKF_CUSTOM_DUMP")) {  // debugging aid: the generated source and its code object
    const std::string base = std::string(dir) + "/cdkf_custom_wg_" + std::to_string(bytes) + "_" + std::to_string(ept) + "_" + std::to_string(ukf) +
                             "_" + std::to_string(smoother);
    if (FILE* f = fopen((base + ".hip").c_str(), "w")) {
      fwrite(src.data(), 1, src.size(), f);
      fclose(f);
    }
    if (FILE* f = fopen((base + ".co").c_str(), "wb")) {
      fwrite(code.data(), 1, code.size(), f);
      fclose(f);
    }
  }
  return CDKF_OK;
}

int get_wg_function(int kind, const WgKey& key, hipFunction_t* fn) {
  std::lock_guard<std::mutex> lock(g_mutex);
  int dev = 0;
  CDKF_HIP_CHECK(hipGetDevice(&dev));
  auto it = g_wg_modules.find({dev, key});
  if (it != g_wg_modules.end()) {
    *fn = it->second.fn;
    return CDKF_OK;
  }
  const CustomDrift& c = g_drifts[kind - CDKF_DRIFT_CUSTOM_BASE];
  hipDeviceProp_t prop;
  CDKF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  std::vector<char> code;
  std::string lowered;
  int rc = compile_wg_variant(c, key, prop.gcnArchName, code, lowered);
  if (rc) return rc;
  Compiled m;
  CDKF_HIP_CHECK(hipModuleLoadData(&m.module, code.data()));
  CDKF_HIP_CHECK(hipModuleGetFunction(&m.fn, m.module, lowered.c_str()));
  g_wg_modules[{dev, key}] = m;
  *fn = m.fn;
  return CDKF_OK;
}

}  // namespace

// the registered parameter count of a custom drift of this state dimension (-1: no such drift): what launch_wg.hip checks n_theta against
long custom_ntheta(int kind, int state_dim) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const int idx = kind - CDKF_DRIFT_CUSTOM_BASE;
  if (idx < 0 || idx >= (int)g_drifts.size() || g_drifts[idx].d != state_dim) return -1;
  return g_drifts[idx].n_theta;
}

// launch of the (filter, smoother) pair of launch_wg_dispatch for a drift that was given as source
template <typename R>
int launch_custom_wg(const WgArgs<R>& a, int ept, bool filter, bool smoother, int threads, size_t lds_f, size_t lds_s, hipStream_t stream) {
  auto run = [&](int smooth, size_t lds) -> int {
    hipFunction_t fn = nullptr;
    const size_t bytes = (lds + 15) & ~size_t(15);
    int r = get_wg_function(a.kind, WgKey(a.kind, (int)sizeof(R), ept, smooth ? 0 : a.ukf, smooth, (long)bytes, a.du), &fn);
    if (r) return r;
    WgArgs<R> arg = a;
    void* args[] = {(void*)&arg};
    note_kernel(smooth ? "ekf_smoother_wg_kernel<%s, %d> (custom drift)" : "ekf_filter_wg_kernel<%s, %d, ...> (custom drift)", real_name<R>(), ept);
    CDKF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)a.N, 1, 1, (unsigned)threads, 1, 1, 0, stream, args, nullptr));
    return CDKF_OK;
  };
  int rc = CDKF_OK;
  if (filter) rc = run(0, lds_f);
  if (!rc && smoother) rc = run(1, lds_s);
  return rc;
}
// ... and of the reverse sweep (launch_adjwg.hip)
template <typename R>
int launch_custom_awg(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, long scratch_stride, int cap, int ne, size_t lds, hipStream_t stream) {
  hipFunction_t fn = nullptr;
  const size_t bytes = (lds + 15) & ~size_t(15);
  int r = get_wg_function(a.kind, WgKey(a.kind, (int)sizeof(R), ne, 0, 2, (long)bytes, a.du), &fn);
  if (r) return r;
  WgArgs<R> arg = a;
  void* args[] = {(void*)&arg, (void*)&grad, (void*)&grad_model, (void*)&scratch, (void*)&scratch_stride, (void*)&cap};
  note_kernel("ekf_adjoint_wg_kernel<%s, %d> (custom drift)", real_name<R>(), ne);
  CDKF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)a.N, 1, 1, 256, 1, 1, 0, stream, args, nullptr));
  return CDKF_OK;
}
template int launch_custom_awg<float>(const WgArgs<float>&, float*, float*, float*, long, int, int, size_t, hipStream_t);
template int launch_custom_awg<double>(const WgArgs<double>&, double*, double*, double*, long, int, int, size_t, hipStream_t);

// the reverse sweep takes a drift given as source if: linear emission, state_order 'first' or 'second' (grad(div f) registered as
// identically zero or as "auto"), and the (column, direction) tasks of its second-derivative contraction fit the workgroup: state_dim + n_theta <= 256
// and <= one LDS slot (q x ld reals, q = max(state_dim, emission_dim)) or 64
bool custom_adjoint_available(const cdkf_model* mdl, const cdkf_opts* o) {
  if (!custom_kind(mdl->drift_kind) || mdl->emission_kind != 0) return false;
  if (o->num_iter != 1 || o->forecast || o->state_order == CDKF_ORDER_ZEROTH) return false;
  std::lock_guard<std::mutex> lock(g_mutex);
  const CustomDrift& c = g_drifts[mdl->drift_kind - CDKF_DRIFT_CUSTOM_BASE];
  if (c.d != mdl->state_dim || c.n_theta != mdl->n_theta) return false;
  // 'second': grad(div f) registered as identically zero, or "auto" (third derivatives by triply nested dual numbers)
  if (o->state_order == CDKF_ORDER_SECOND && !(c.has_g && (c.auto_g || blank(c.g_src)))) return false;
  const int q = c.d > mdl->emission_dim ? c.d : mdl->emission_dim, Z = c.d + c.n_theta;
  return Z <= 256 && (Z <= q * (q | 1) || Z <= 64);  // (the smallest shapes: the partial sums meet in a 64-entry vector instead of a slot)
}

template int launch_custom_wg<float>(const WgArgs<float>&, int, bool, bool, int, size_t, size_t, hipStream_t);
template int launch_custom_wg<double>(const WgArgs<double>&, int, bool, bool, int, size_t, size_t, hipStream_t);

bool custom_emission_kind(int ek, int d, int m) {
  std::lock_guard<std::mutex> lock(g_mutex_emis());
  const int idx = ek - CDKF_EMISSION_CUSTOM_BASE;
  return idx >= 0 && idx < (int)g_emis.size() && g_emis[idx].d == d && g_emis[idx].m == m;
}

static int c_dim(int kind) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const int idx = kind - CDKF_DRIFT_CUSTOM_BASE;
  return (idx >= 0 && idx < (int)g_drifts.size()) ? g_drifts[idx].d : -1;
}

int custom_emission_register(int state_dim, int emission_dim, const char* h_src, const char* hjac_src) {
  // (up to six dimensions: the register-resident kernels; above, up to sixteen: the value mode of the tangent sweeps -- filters,
  //  log-likelihood gradients; the smoother's backward sweep and the forecasts on the workgroup kernels: launch_custom's dispatch)
  if (state_dim < 1 || state_dim > 16 || emission_dim < 1 || emission_dim > 16 || !h_src) {
    set_error("custom emission: need 1 <= state_dim, emission_dim <= 16 and the source of h (hjac_src NULL or empty: its Jacobian is "
              "derived from h_src by dual numbers)");
    return CDKF_EINVAL;
  }
  if (!hjac_src) hjac_src = "";
  std::lock_guard<std::mutex> lock(g_mutex_emis());
  for (size_t k = 0; k < g_emis.size(); ++k)
    if (g_emis[k].d == state_dim && g_emis[k].m == emission_dim && g_emis[k].h_src == h_src && g_emis[k].jac_src == hjac_src)
      return CDKF_EMISSION_CUSTOM_BASE + (int)k;
  g_emis.push_back(CustomEmission{state_dim, emission_dim, h_src, hjac_src});
  return CDKF_EMISSION_CUSTOM_BASE + (int)g_emis.size() - 1;
}

bool custom_kind(int kind) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return kind >= CDKF_DRIFT_CUSTOM_BASE && kind - CDKF_DRIFT_CUSTOM_BASE < (int)g_drifts.size();
}

bool custom_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  if (!custom_kind(mdl->drift_kind)) return false;
  int cd = 0;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    const CustomDrift& c = g_drifts[mdl->drift_kind - CDKF_DRIFT_CUSTOM_BASE];
    if (c.d != mdl->state_dim || c.n_theta != mdl->n_theta) return false;
    if (o && o->state_order == CDKF_ORDER_SECOND && !c.has_g) return false;
    cd = c.d;
  }
  if (cd > 6 || mdl->emission_dim > 6) {  // the workgroup kernels: linear emission, their LDS plan (asked in fp32; an fp64 launch that
    if (mdl->emission_kind == 0) return custom_wg_fits(mdl);  // does not fit says so itself)
    if (o && o->forecast) {  // (forecasts never evaluate the emission: the workgroup kernels, as if it were linear)
      cdkf_model lin = *mdl;
      lin.emission_kind = 0;
      return custom_wg_fits(&lin);
    }
    return o && ukf_tangent_available(mdl, o);  // (value mode of the tangent sweeps; the extended filter's own conditions: at the launch)
  }
  return true;
}

// the log-likelihood gradient w.r.t. theta (forward sensitivities, a lane per (trajectory, parameter): ekf_grad_reg_body with the
// drift's parameter derivatives from dual numbers): linear emission, num_iter 1, state_order 'first' -- or 'second' with a grad(div f)
// that was registered as identically zero (an empty divgrad_src): the sweep does not carry the mean's second-order term
bool custom_grad_available(const cdkf_model* mdl, const cdkf_opts* o) {
  if (!custom_shape_available(mdl, o) || mdl->emission_kind != 0 || mdl->n_theta < 1) return false;
  if (mdl->state_dim > 6 || mdl->emission_dim > 6) return false;  // (the forward-sensitivity sweep is a register-resident kernel)
  if (o->num_iter != 1 || o->forecast || o->state_order == CDKF_ORDER_ZEROTH) return false;
  std::lock_guard<std::mutex> lock(g_mutex);
  const CustomDrift& c = g_drifts[mdl->drift_kind - CDKF_DRIFT_CUSTOM_BASE];
  return o->state_order == CDKF_ORDER_FIRST || (c.has_g && !c.auto_g && blank(c.g_src));
}

// The two argument blocks of a run-time compiled register-resident kernel (the generated `unpack` reads them back):
//   reals: theta | LQL | LQLz | H | hb | Rm | m0 | P0 | dt0 dt_final ukf_c ukf_wm0 ukf_wc0 ukf_wi | rk.a[30] rk.b[6] rk.berr[7] rtol atol c1 c2 c3
//          | dtmin dtmax safety factormin factormax
//   longs: max_steps order num_iter forecast N T t_sn t_sk y_sn y_sk y_si m_sn m_sk m_si P_sn P_sk P_si stages solver adaptive fsal lanes xcd_shift
//          u_sn u_sk u_si
// Host arithmetic only (cdkf_debug_custom_reg_blob hands them to the CPU-sanitizer build of the same kernel, tests/test_hostsim.py).
// (round 5: dtmin / dtmax were missing -- an adaptive solve of a source drift on these kernels clipped its steps to whatever the
//  uninitialised fields of the kernel's argument struct held; found by reading the generated source for the MemorySanitizer build)
template <typename R>
static RegGrouping custom_reg_blob(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, bool smoother, bool gradient, bool no_y,
                                   std::vector<R>& par, long (&ip)[26]) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  const int np = d * (d + 1) / 2;
  par.clear();
  for (long k = 0; k < mdl->n_theta; ++k) par.push_back(R(mdl->theta[k]));
  std::vector<R> packed(np);
  lql_packed<R>(mdl->L, mdl->Qc, d, 1.0, packed.data());
  par.insert(par.end(), packed.begin(), packed.end());
  lql_packed<R>(mdl->L, mdl->Qc, d, o->cov_rescaling, packed.data());
  par.insert(par.end(), packed.begin(), packed.end());
  for (int k = 0; k < m * d; ++k) par.push_back(R(mdl->H[k]));
  for (int k = 0; k < m; ++k) par.push_back(R(mdl->h_bias[k]));
  for (int k = 0; k < m * m; ++k) par.push_back(R(mdl->R[k]));
  for (int k = 0; k < d; ++k) par.push_back(R(mdl->m0[k]));
  for (int i = 0; i < d; ++i)
    for (int j = i; j < d; ++j) par.push_back(R(0.5) * (R(mdl->P0[i * d + j]) + R(mdl->P0[j * d + i])));
  {
    const R alpha = R(o->ukf_alpha), n = R(d);
    const R lamb = alpha * alpha * (n + R(o->ukf_kappa)) - n;
    par.push_back(R(o->dt0));
    par.push_back(R(o->dt_final));
    par.push_back(std::sqrt(n + lamb));
    par.push_back(lamb / (n + lamb));
    par.push_back(lamb / (n + lamb) + (R(1) - alpha * alpha + R(o->ukf_beta)));
    par.push_back(R(1) / (R(2) * (n + lamb)));
  }
  RkTab<R> tb;
  fill_rk_tab<R>(o, tb);
  for (int s = 0; s < 6; ++s)
    for (int j = 0; j < 5; ++j) par.push_back(tb.a[s][j]);
  for (int s = 0; s < 6; ++s) par.push_back(tb.b[s]);
  for (int s = 0; s < 7; ++s) par.push_back(tb.berr[s]);
  par.push_back(tb.rtol); par.push_back(tb.atol); par.push_back(tb.c1); par.push_back(tb.c2); par.push_back(tb.c3);
  par.push_back(tb.dtmin); par.push_back(tb.dtmax);
  par.push_back(tb.safety); par.push_back(tb.fmin); par.push_back(tb.fmax);
  const RegGrouping grouping = reg_grouping(gradient ? N * mdl->n_theta : N, (int)sizeof(R));  // (gradient: a lane per (trajectory, parameter))
  ip[21] = grouping.lanes;
  ip[22] = grouping.xcd_shift;
  ip[17] = tb.stages;
  ip[18] = o->solver;
  ip[19] = tb.adaptive;
  ip[20] = tb.fsal;
  ip[0] = (long)o->max_steps;
  ip[1] = o->state_order;
  ip[2] = smoother ? 1 : o->num_iter;
  ip[3] = smoother ? 0 : o->forecast;
  ip[4] = N;
  ip[5] = T;
  {
    const SweepStrides ss = sweep_strides(o, N, T, d, m, no_y);
    long* st = ip + 6;  // t_sn t_sk y_sn y_sk y_si m_sn m_sk m_si P_sn P_sk P_si
    st[0] = ss.t_sn; st[1] = ss.t_sk; st[2] = ss.y_sn; st[3] = ss.y_sk; st[4] = ss.y_si; st[5] = ss.m_sn; st[6] = ss.m_sk;
    st[7] = ss.m_si; st[8] = ss.P_sn; st[9] = ss.P_sk; st[10] = ss.P_si;
    const int lin = (o->layout_in == CDKF_LAYOUT_SAME) ? o->layout : o->layout_in;
    const ArrayStrides us = layout_strides(lin, N, T, mdl->input_dim > 0 ? mdl->input_dim : 1);
    ip[23] = us.sn; ip[24] = us.sk; ip[25] = us.si;
  }
  return grouping;
}

template <typename R>
static int launch_tangent_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm, R* fc,
                                 R* pm, R* pc, int32_t* status, hipStream_t stream, bool ekf);
template <typename R>  // (launch_wg.hip)
int launch_ekf_smoother_backward_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm,
                                    R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream);

// algo: 0 EKF filter, 1 UKF filter, 2 EKF smoother (filter + backward sweep), 3 EKF log-likelihood + gradient (a1: grad [N, n_theta])
template <typename R>
int launch_custom(int algo, const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                  R* a1, R* a2, R* a3, R* a4, int32_t* status, hipStream_t stream) {
  if (!custom_kind(mdl->drift_kind)) {
    set_error("unknown drift_kind %d", mdl->drift_kind);
    return CDKF_EUNSUPPORTED;
  }
  CustomDrift c;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    c = g_drifts[mdl->drift_kind - CDKF_DRIFT_CUSTOM_BASE];
  }
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (c.d != d || c.n_theta != mdl->n_theta) {
    set_error("custom drift %d was registered for state_dim=%d, n_theta=%d (model has %d, %lld)", mdl->drift_kind, c.d,
              c.n_theta, d, (long long)mdl->n_theta);
    return CDKF_EINVAL;
  }
  const int ek = mdl->emission_kind;
  if (d > 6 || m > 6) {  // beyond the register-resident kernels: the workgroup-per-trajectory sweeps with this drift compiled in
    if (ek) {  // an emission given as source above six dimensions: the literal recursions of cdkf_ukf_tangent_kernels.h in value mode
      if (o->forecast && (algo == 0 || algo == 1)) {  // repeated _predict (inference_ekf.py:679-766, inference_ukf.py:409-505): no emission in it
        cdkf_model lin = *mdl;
        lin.emission_kind = 0;
        if (!custom_wg_fits(&lin)) {
          set_error("forecast with a custom emission above six dimensions: state_dim %d, emission_dim %d do not fit the workgroup kernels' LDS plan", d, m);
          return CDKF_EUNSUPPORTED;
        }
        if (algo == 0 && o->state_order == CDKF_ORDER_SECOND && !c.has_g) {
          set_error("custom drift without grad(div f): state_order 'second' needs it; register divgrad_src \"auto\" or use state_order 'first'");
          return CDKF_EUNSUPPORTED;
        }
        return algo == 0 ? launch_ekf_filter_wg<R>(&lin, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream)
                         : launch_ukf_filter_wg<R>(&lin, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream);
      }
      if (algo == 0 || algo == 1) return launch_tangent_filter<R>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream, algo == 0);
      if (algo == 2) {  // the smoother: that forward pass (num_iter 1, inference_ekf.py:489-495), then the workgroup kernels' backward sweep,
        if (!a1 || !a2 || !a3 || !a4) {  // which reads the filtered moments and the drift only
          set_error("EKF smoother: filtered and smoothed output pointers must not be NULL");
          return CDKF_EINVAL;
        }
        cdkf_model lin = *mdl;
        lin.emission_kind = 0;
        if (!custom_wg_fits(&lin)) {
          set_error("EKF smoother with a custom emission above six dimensions: state_dim %d, emission_dim %d do not fit the workgroup "
                    "kernels' LDS plan (the backward sweep runs there)", d, m);
          return CDKF_EUNSUPPORTED;
        }
        cdkf_opts o1 = *o;
        o1.num_iter = 1;
        const int rc = launch_tangent_filter<R>(mdl, &o1, N, T, t, y, ll, a1, a2, nullptr, nullptr, status, stream, true);
        return rc ? rc : launch_ekf_smoother_backward_wg<R>(&lin, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream);
      }
      set_error("custom emissions above six dimensions: filters, smoother and the log-likelihood gradients of the tangent kernels "
                "(the forward-sensitivity sweep is a register-resident kernel, state_dim, emission_dim <= 6; got %d, %d)", d, m);
      return CDKF_EUNSUPPORTED;
    }
    if (algo == 3) {  // (launch_ekf_grad sends these shapes to the reverse sweep, not here)
      set_error("custom drift: the forward-sensitivity sweep is a register-resident kernel, state_dim, emission_dim <= 6 (got %d, %d)", d, m);
      return CDKF_EUNSUPPORTED;
    }
    if (algo != 1 && o->state_order == CDKF_ORDER_SECOND && !c.has_g) {
      set_error("custom drift without grad(div f): state_order 'second' needs it (the reference differentiates the drift twice, "
                "inference_ekf.py:108-116); register divgrad_src \"auto\" or use state_order 'first'");
      return CDKF_EUNSUPPORTED;
    }
    if (algo == 0) return launch_ekf_filter_wg<R>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream);
    if (algo == 1) return launch_ukf_filter_wg<R>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream);
    return launch_ekf_smoother_wg<R>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status, stream);
  }
  if (ek && !custom_emission_kind(ek, d, m)) {
    set_error("emission_kind %d is not a custom emission registered for state_dim=%d, emission_dim=%d", ek, d, m);
    return CDKF_EINVAL;
  }
  if (algo != 1 && o->state_order == CDKF_ORDER_SECOND && !c.has_g) {
    set_error("custom drift without grad(div f) source: state_order 'second' needs it (the reference differentiates the "
              "drift twice, inference_ekf.py:108-116); register divgrad_src or use state_order 'first'");
    return CDKF_EUNSUPPORTED;
  }
  const bool smoother = algo == 2, gradient = algo == 3;
  if (gradient && (!custom_grad_available(mdl, o) || !a1 || !y)) {
    set_error("custom drift: the gradient sweep needs a linear emission, n_theta >= 1, num_iter 1 and state_order 'first' (or "
              "'second' with grad(div f) registered as identically zero)");
    return CDKF_EUNSUPPORTED;
  }
  if (smoother && (!a1 || !a2 || !a3 || !a4)) {
    set_error("EKF smoother: filtered and smoothed output pointers must not be NULL");
    return CDKF_EINVAL;
  }
  // ---- parameter blobs -----------------------------------------------------------------------------------------------
  std::vector<R> par;
  long ip[26];
  const RegGrouping grouping = custom_reg_blob<R>(mdl, o, N, T, smoother, gradient, !y, par, ip);
  const int du = mdl->input_dim;
  const R* uu = du > 0 ? (const R*)o->inputs : nullptr;  // (device memory here: the host entry points have uploaded it)
  const int generic = o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive;
  const R* yy = y ? y : t;  // forecast mode ignores the observations; keep the prefetch loads on valid memory
  const size_t par_bytes = par.size() * sizeof(R), blob = ((par_bytes + 15) & ~size_t(15)) + sizeof(ip);
  ParamLease lease(stream);
  int rc = param_pool_acquire(blob, &lease.slot);
  if (rc) return rc;
  ParamSlot* slot = lease.slot;
  std::memcpy(slot->host, par.data(), par_bytes);
  std::memcpy((char*)slot->host + ((par_bytes + 15) & ~size_t(15)), ip, sizeof(ip));
  CDKF_HIP_CHECK(hipMemcpyAsync(slot->dev, slot->host, blob, hipMemcpyHostToDevice, stream));
  const R* dpar = (const R*)slot->dev;
  const long* dip = (const long*)((char*)slot->dev + ((par_bytes + 15) & ~size_t(15)));

  // ---- kernels ---------------------------------------------------------------------------------------------------------
  const unsigned blocks = grouping.blocks;
  R* null_r = nullptr;
  auto run = [&](const Key& key, R* o1, R* o2, R* o3, R* o4, R* sm, R* sP) -> int {
    hipFunction_t fn = nullptr;
    int r = get_function(mdl->drift_kind, key, &fn);
    if (r) return r;
    void* args[] = {(void*)&dpar, (void*)&dip, (void*)&t, (void*)&yy, (void*)&ll, (void*)&o1, (void*)&o2,
                    (void*)&o3,   (void*)&o4,  (void*)&status, (void*)&sm, (void*)&sP, (void*)&uu};
    CDKF_HIP_CHECK(hipModuleLaunchKernel(fn, blocks, 1, 1, 64, 1, 1, 0, stream, args, nullptr));
    return CDKF_OK;
  };
  const int zeroth = (algo != 1 && o->state_order == CDKF_ORDER_ZEROTH) ? 1 : 0;
  if (gradient) {
    rc = run(Key(mdl->drift_kind, (int)sizeof(R), m, 0, 0, 0, 2, generic, 0, du), a1, null_r, null_r, null_r, null_r, null_r);
  } else if (!smoother) {
    rc = run(Key(mdl->drift_kind, (int)sizeof(R), m, algo == 1, zeroth, o->forecast ? 1 : 0, 0, generic, ek, du), a1, a2, a3, a4, null_r, null_r);
  } else {
    rc = run(Key(mdl->drift_kind, (int)sizeof(R), m, 0, zeroth, 0, 0, generic, ek, du), a1, a2, null_r, null_r, null_r, null_r);
    if (!rc) rc = run(Key(mdl->drift_kind, (int)sizeof(R), m, 0, 0, 0, 1, generic, 0, du), a1, a2, null_r, null_r, a3, a4);
  }
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

template int launch_custom<float>(int, const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, float*,
                                  float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_custom<double>(int, const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*,
                                   double*, double*, double*, double*, double*, int32_t*, hipStream_t);

int custom_register(int state_dim, int n_theta, const char* f_src, const char* jac_src, const char* divgrad_src) {
  if (state_dim < 1 || state_dim > 64 || n_theta < 0 || !f_src) {
    set_error("custom drift: need 1 <= state_dim <= 64, n_theta >= 0 and the source of f (jac_src NULL or empty: the Jacobian is "
              "derived from f_src by dual numbers; divgrad_src \"auto\": so is grad(div f))");
    return CDKF_EINVAL;
  }
  const bool auto_jac = !jac_src || blank(jac_src);
  const bool auto_g = divgrad_src && std::string(divgrad_src) == "auto";
  if (state_dim > 6 && (!auto_jac || (divgrad_src && !auto_g && !blank(divgrad_src)))) {
    // (a thread of the workgroup kernels evaluates ONE direction of the Jacobian; a source that fills all of F[D][D] has no place there)
    set_error("custom drift: above state_dim 6 the derivatives come from f_src by dual numbers -- pass jac_src NULL and divgrad_src "
              "NULL, \"\" (identically zero) or \"auto\" (got state_dim %d)", state_dim);
    return CDKF_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  CustomDrift c{state_dim, n_theta, f_src, auto_jac ? "" : jac_src, (divgrad_src && !auto_g) ? divgrad_src : "", divgrad_src != nullptr,
                auto_jac, auto_g};
  for (size_t k = 0; k < g_drifts.size(); ++k) {
    const CustomDrift& e = g_drifts[k];
    if (e.d == c.d && e.n_theta == c.n_theta && e.f_src == c.f_src && e.jac_src == c.jac_src && e.g_src == c.g_src &&
        e.has_g == c.has_g && e.auto_jac == c.auto_jac && e.auto_g == c.auto_g)
      return CDKF_DRIFT_CUSTOM_BASE + (int)k;
  }
  g_drifts.push_back(c);
  return CDKF_DRIFT_CUSTOM_BASE + (int)g_drifts.size() - 1;
}

int custom_compile_check(int kind, int bytes_per_real, int emission_dim, int algo, int state_order, int emission_kind) {
  // algo + 16: the register-resident variant that reads its Runge-Kutta tableau / step-size controller from the arguments (what a
  // launch with opts.solver != Dormand-Prince or opts.adaptive compiles) instead of the pinned Dormand-Prince constants
  // ... + 256 * input_dim: the variant for a model with that many inputs per row (cdkf_model.input_dim; the snippets' `u`)
  const int du = algo / 256;
  algo -= 256 * du;
  const int generic = (algo >= 16) ? 1 : 0;
  if (generic) algo -= 16;
  if (du < 0 || du > 64) {
    set_error("custom drift compile check: input_dim %d", du);
    return CDKF_EINVAL;
  }
  if (emission_kind && !custom_emission_kind(emission_kind, c_dim(kind), emission_dim)) {
    set_error("custom emission %d is not registered for this state / emission dimension", emission_kind);
    return CDKF_EINVAL;
  }
  if (!custom_kind(kind) || (bytes_per_real != 4 && bytes_per_real != 8) || emission_dim < 1 || emission_dim > 64 || algo < 0 ||
      algo > 3) {
    set_error("custom drift compile check: bad arguments");
    return CDKF_EINVAL;
  }
  CustomDrift c;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    c = g_drifts[kind - CDKF_DRIFT_CUSTOM_BASE];
  }
  std::vector<char> code;
  if (c.d > 6 || emission_dim > 6) {  // the workgroup kernels (dense emission matrix assumed: the larger LDS plan)
    if (emission_kind) {
      set_error("custom drift compile check: above six dimensions a custom emission runs on the tangent kernels' value mode -- "
                "cdkf_ukf_tangent_compile / cdkf_ekf_tangent_compile check those");
      return CDKF_EUNSUPPORTED;
    }
    if (algo == 3) {  // the reverse sweep (the forward sweep is algo 0's kernel)
      int ne = 0;
      size_t lds = 0;
      if (custom_awg_geometry(c.d, emission_dim, bytes_per_real, &ne, &lds)) {
        set_error("custom drift compile check: state_dim %d, emission_dim %d do not fit the reverse sweep's LDS plan", c.d, emission_dim);
        return CDKF_EUNSUPPORTED;
      }
      std::string lowered;
      return compile_wg_variant(c, WgKey(kind, bytes_per_real, ne, 0, 2, (long)((lds + 15) & ~size_t(15)), du), "gfx950", code, lowered);
    }
    int ept = 0, threads = 0;
    size_t lds_f = 0, lds_s = 0;
    if (custom_wg_geometry(kind, c.d, emission_dim, bytes_per_real, algo == 1, &ept, &threads, &lds_f, &lds_s)) {
      set_error("custom drift compile check: state_dim %d, emission_dim %d do not fit the workgroup kernels' LDS plan", c.d, emission_dim);
      return CDKF_EUNSUPPORTED;
    }
    std::string lowered;
    int rc = compile_wg_variant(c, WgKey(kind, bytes_per_real, ept, algo == 1, 0, (long)((lds_f + 15) & ~size_t(15)), du), "gfx950", code, lowered);
    if (!rc && algo == 2) rc = compile_wg_variant(c, WgKey(kind, bytes_per_real, ept, 0, 1, (long)((lds_s + 15) & ~size_t(15)), du), "gfx950", code, lowered);
    return rc;
  }
  const int zeroth = (algo != 1 && state_order == CDKF_ORDER_ZEROTH) ? 1 : 0;
  if (algo == 3) return compile_variant(c, Key(kind, bytes_per_real, emission_dim, 0, 0, 0, 2, generic, 0, du), "gfx950", code);
  int rc = compile_variant(c, Key(kind, bytes_per_real, emission_dim, algo == 1, zeroth, 0, 0, generic, emission_kind, du), "gfx950", code);
  if (!rc && algo == 2) rc = compile_variant(c, Key(kind, bytes_per_real, emission_dim, 0, 0, 0, 1, generic, 0, du), "gfx950", code);
  return rc;
}

int rtc_exec_prologue_check(const std::vector<char>& code, const std::string& arch, std::string* where) {
  return rtc_exec_prologue_defect(code, arch, where);
}

// ---- the unscented filter's gradient for any drift / emission: forward mode through the literal recursion (cdkf_ukf_tangent_kernels.h) ----
// The drift's statements: the registered source, or -- for a built-in drift -- the same function written out here (theta layouts:
// cdkf.h).  The emission's: the registered source, or the linear emission h = H x + b over eta = [H | b].
static std::string ut_builtin_drift(const cdkf_model* mdl) {
  const int d = mdl->state_dim;
  const std::string D_ = std::to_string(d);
  switch (mdl->drift_kind) {
    case CDKF_DRIFT_LINEAR:
      return "for (int i_ = 0; i_ < " + D_ + "; ++i_) { T s_ = theta[" + std::to_string(d * d) + " + i_]; for (int k_ = 0; k_ < " + D_ +
             "; ++k_) s_ += theta[i_ * " + D_ + " + k_] * x[k_]; fx[i_] = s_; }";
    case CDKF_DRIFT_LORENZ63:
      return "fx[0] = theta[0] * (x[1] - x[0]); fx[1] = x[0] * (theta[1] - x[2]) - x[1]; fx[2] = x[0] * x[1] - theta[2] * x[2];";
    case CDKF_DRIFT_LORENZ96:
      return "for (int i_ = 0; i_ < " + D_ + "; ++i_) fx[i_] = (x[(i_ + 1) % " + D_ + "] - x[(i_ + " + std::to_string(d - 2) + ") % " + D_ +
             "]) * x[(i_ + " + std::to_string(d - 1) + ") % " + D_ + "] - x[i_] + theta[0];";
    case CDKF_DRIFT_MLP_TANH: {
      const int h1 = mdl->hidden1, h2 = mdl->hidden2;
      const int oW1 = 0, ob1 = oW1 + h1 * d, oW2 = ob1 + h1, ob2 = oW2 + h2 * h1, oW3 = ob2 + h2, ob3 = oW3 + d * h2;
      auto S = [](int v) { return std::to_string(v); };
      return "T a1_[" + S(h1) + "], a2_[" + S(h2) + "];\n"
             "for (int i_ = 0; i_ < " + S(h1) + "; ++i_) { T s_ = theta[" + S(ob1) + " + i_]; for (int k_ = 0; k_ < " + D_ + "; ++k_) s_ += theta[" + S(oW1) +
             " + i_ * " + D_ + " + k_] * x[k_]; a1_[i_] = tanh(s_); }\n"
             "for (int i_ = 0; i_ < " + S(h2) + "; ++i_) { T s_ = theta[" + S(ob2) + " + i_]; for (int k_ = 0; k_ < " + S(h1) + "; ++k_) s_ += theta[" + S(oW2) +
             " + i_ * " + S(h1) + " + k_] * a1_[k_]; a2_[i_] = tanh(s_); }\n"
             "for (int i_ = 0; i_ < " + D_ + "; ++i_) { T s_ = theta[" + S(ob3) + " + i_]; for (int k_ = 0; k_ < " + S(h2) + "; ++k_) s_ += theta[" + S(oW3) +
             " + i_ * " + S(h2) + " + k_] * a2_[k_]; fx[i_] = s_; }";
    }
    default:
      return std::string();
  }
}

constexpr int kUtMaxDim = 16;  // state / emission dimension of the tangent sweep (private arrays: ~ 6 (d + d (d + 1) / 2) dual numbers per lane)

// is the model one the tangent sweep takes?  (`why`: the refusal's text)
static bool ut_model_sources(const cdkf_model* mdl, const cdkf_opts* o, std::string& f_src, std::string& h_src, int& nth, std::string* why, bool ekf = false) {
  auto no = [&](const char* msg) {
    if (why) *why = msg;
    return false;
  };
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (d < 1 || m < 1 || d > kUtMaxDim || m > kUtMaxDim) return no("state_dim, emission_dim <= 16");
  if (o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive) return no("the default solver (fixed-step Dormand-Prince)");
  if (o->forecast) return no("no forecast mode");
  if (ekf) {  // the extended filter's tangent sweep: jacfwd by an outer dual level, grad(div f) by two (state_dim <= 8)
    if (o->state_order == CDKF_ORDER_ZEROTH) return no("state_order first or second");
    if (o->state_order == CDKF_ORDER_SECOND && d > 8) return no("state_order first above eight dimensions (second: state_dim <= 8)");
    if (o->num_iter < 1 || o->num_iter > 64) return no("1 <= num_iter <= 64");
  }
  nth = (int)mdl->n_theta;
  if (custom_kind(mdl->drift_kind)) {
    std::lock_guard<std::mutex> lock(g_mutex);
    const CustomDrift& c = g_drifts[mdl->drift_kind - CDKF_DRIFT_CUSTOM_BASE];
    if (c.d != d || c.n_theta != mdl->n_theta) return no("the dimensions the drift was registered with");
    if (ekf && o->state_order == CDKF_ORDER_SECOND && !c.has_g) return no("grad(div f) registered for state_order second (divgrad_src, as the filter asks)");
    f_src = c.f_src;
  } else {
    f_src = ut_builtin_drift(mdl);
    if (f_src.empty()) return no("a drift whose statements the tangent sweep has (linear, Lorenz-63, Lorenz-96, MLP, or source)");
    if (mdl->drift_kind == CDKF_DRIFT_LORENZ96 && d < 4) return no("Lorenz-96 from four dimensions");
  }
  if (mdl->emission_kind) {
    if (!custom_emission_kind(mdl->emission_kind, d, m)) return no("an emission registered for these dimensions");
    std::lock_guard<std::mutex> lock(g_mutex_emis());
    h_src = g_emis[mdl->emission_kind - CDKF_EMISSION_CUSTOM_BASE].h_src;
  } else {
    h_src = "for (int r_ = 0; r_ < " + std::to_string(m) + "; ++r_) { T s_ = eta[" + std::to_string(m * d) + " + r_]; for (int k_ = 0; k_ < " +
            std::to_string(d) + "; ++k_) s_ += eta[r_ * " + std::to_string(d) + " + k_] * x[k_]; hx[r_] = s_; }";
  }
  return true;
}

bool ukf_tangent_available(const cdkf_model* mdl, const cdkf_opts* o) {
  std::string f, h;
  int nth = 0;
  return ut_model_sources(mdl, o, f, h, nth, nullptr);
}
bool ekf_tangent_available(const cdkf_model* mdl, const cdkf_opts* o) {
  std::string f, h;
  int nth = 0;
  return ut_model_sources(mdl, o, f, h, nth, nullptr, true);
}

// The emission moments of Gaussian state marginals under an emission given as SOURCE (emissions_extended_kalman_filter,
// inference_ekf.py:768-855: h(m), jacfwd(h) P jacfwd(h)^T + R; emissions_unscented_kalman_filter, inference_ukf.py:507-612: the sigma
// points of (m, P) through h, weighted mean and covariance + R): a lane per (m, P) row.  Kept as TEXT of the generated translation unit
// (it is hashed into the code-object cache key with the model's statements), after `using R = ...` and the model struct.
static const char* kEmissionMomentsKernel = R"EM(
struct EmArgs {
  const R* par;  // eta = H [M, D] | h_bias [M], then R [M, M]
  const R* t;    // [rows] or null
  const R* u;    // [rows, DU] or null
  const R* mu;   // [rows, D]
  const R* P;    // [rows, D, D] or null: point estimates, means only
  R* ym;         // [rows, M]
  R* yc;         // [rows, M, M] or null
  long rows;
  R c, wm0, wc0, wi;  // sqrt(D + lambda), lambda / (D + lambda), wm0 + 1 - alpha^2 + beta, 1 / (2 (D + lambda))
  int ukf;
};
namespace cdkf {
template <typename T>
struct EmEta {
  const R* v;
  __device__ T operator[](int k) const { return T(v[k]); }
};
}  // namespace cdkf
extern "C" __global__ __launch_bounds__(64) void cdkf_emission_moments_kernel(const EmArgs a) {
  using namespace cdkf;
  typedef UtModel MD;
  constexpr int D = MD::D, M = MD::M, DU = MD::DU;
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.rows) return;
  const R* eta = a.par;
  const R* Rm = a.par + M * D + M;
  R ub[DU > 0 ? DU : 1];
  ub[0] = R(0);
  for (int i = 0; i < DU; ++i) ub[i] = a.u ? a.u[r * DU + i] : R(0);
  const R tt = a.t ? a.t[r] : R(0);
  const R* m = a.mu + r * D;
  R* ym = a.ym + r * M;
  if (!a.P || !a.yc) {  // point estimates: pushed through h
    R hx[M];
    MD::template h<R, R>(m, EmEta<R>{eta}, hx, ub, tt);
    for (int k = 0; k < M; ++k) ym[k] = hx[k];
    return;
  }
  const R* P = a.P + r * D * D;
  R* yc = a.yc + r * M * M;
  if (!a.ukf) {
    typedef Dual<R, D> J;
    J x[D], hx[M];
    for (int i = 0; i < D; ++i) {
      x[i] = J(m[i]);
      x[i].g[i] = R(1);
    }
    MD::template h<R, J>(x, EmEta<J>{eta}, hx, ub, tt);
    for (int k = 0; k < M; ++k) ym[k] = hx[k].v;
    for (int p = 0; p < M; ++p)
      for (int q = 0; q < M; ++q) {
        R s = Rm[p * M + q];
        for (int i = 0; i < D; ++i) {
          R w = R(0);
          for (int j = 0; j < D; ++j) w += P[i * D + j] * hx[q].g[j];
          s += hx[p].g[i] * w;
        }
        yc[p * M + q] = s;
      }
    return;
  }
  R L[D * D];  // lower Cholesky factor of sym(P) (jnp.linalg.cholesky symmetrises; a non-positive pivot yields NaN)
  for (int j = 0; j < D; ++j) {
    R s = P[j * D + j];
    for (int k = 0; k < j; ++k) s -= L[j * D + k] * L[j * D + k];
    const R piv = sqrt(s);
    L[j * D + j] = piv;
    for (int i = j + 1; i < D; ++i) {
      R w = R(0.5) * (P[i * D + j] + P[j * D + i]);
      for (int k = 0; k < j; ++k) w -= L[i * D + k] * L[j * D + k];
      L[i * D + j] = w / piv;
    }
  }
  R x[D], Y0[M], Yp[D][M], Ym[D][M];
  MD::template h<R, R>(m, EmEta<R>{eta}, Y0, ub, tt);
  for (int i = 0; i < D; ++i) {
    for (int j = 0; j < D; ++j) x[j] = (j >= i) ? m[j] + a.c * L[j * D + i] : m[j];
    MD::template h<R, R>(x, EmEta<R>{eta}, Yp[i], ub, tt);
    for (int j = 0; j < D; ++j) x[j] = (j >= i) ? m[j] - a.c * L[j * D + i] : m[j];
    MD::template h<R, R>(x, EmEta<R>{eta}, Ym[i], ub, tt);
  }
  R yb[M];
  for (int k = 0; k < M; ++k) {
    R s = R(0);
    for (int i = 0; i < D; ++i) s += Yp[i][k] + Ym[i][k];
    yb[k] = a.wm0 * Y0[k] + a.wi * s;
    ym[k] = yb[k];
  }
  for (int p = 0; p < M; ++p)
    for (int q = 0; q < M; ++q) {
      R s = R(0);
      for (int i = 0; i < D; ++i) s += (Yp[i][p] - yb[p]) * (Yp[i][q] - yb[q]) + (Ym[i][p] - yb[p]) * (Ym[i][q] - yb[q]);
      yc[p * M + q] = a.wc0 * (Y0[p] - yb[p]) * (Y0[q] - yb[q]) + a.wi * s + Rm[p * M + q];
    }
}
)EM";

// variant: 0 the unscented filter's tangent sweep, 1 the extended filter's, 2 the emission moments kernel above
static std::string generate_ut_source(const cdkf_model* mdl, const std::string& f_src, const std::string& h_src, int bytes, int variant) {
  const bool ekf = variant == 1;
  const int d = mdl->state_dim, m = mdl->emission_dim, du = mdl->input_dim;
  std::string s;
  s += "#include \"cdkf_ukf_tangent_kernels.h\"\nnamespace cdkf {\nstruct UtModel {\n";
  s += "  static constexpr int D = " + std::to_string(d) + ", M = " + std::to_string(m) + ", NTH = " + std::to_string((int)mdl->n_theta) +
       ", DU = " + std::to_string(du) + ";\n";
  // (theta / eta: views whose operator[] hands out the entry as a T -- the snippets index them, as they index the arrays of the filters)
  s += "  template <typename R, typename T, typename TH> static __device__ void f(const T* x, const TH& theta, T (&fx)[D], const R* u, const R t) {\n"
       "    (void)theta; (void)u; (void)t;\n#line 1 \"drift_f\"\n" + f_src + "\n  }\n";
  s += "  template <typename R, typename T, typename EH> static __device__ void h(const T* x, const EH& eta, T (&hx)[M], const R* u, const R t) {\n"
       "    (void)eta; (void)u; (void)t;\n#line 1 \"emission_h\"\n" + h_src + "\n  }\n};\n}  // namespace cdkf\n";
  s += "using R = " + std::string(bytes == 8 ? "double" : "float") + ";\n";
  if (variant == 2) return s + kEmissionMomentsKernel;
  s += "extern \"C\" __global__ __launch_bounds__(64) void cdkf_ukf_tangent_kernel(const cdkf::UtArgs<R> a) { cdkf::" + std::string(ekf ? "ekf" : "ukf") +
       "_tangent_body<R, cdkf::UtModel>(a); }\n";   // (one entry name for both filters: the harnesses and the launcher call it)
  return s;
}

std::map<std::pair<int, std::string>, Compiled> g_ut_modules;  // (device, cache key)

static int ut_get_function(const cdkf_model* mdl, const cdkf_opts* o, int bytes, hipFunction_t* fn, const char* arch_override, int variant = 0) {
  std::string f_src, h_src, why;
  int nth = 0;
  const bool ekf = variant == 1;
  const char* entry = variant == 2 ? "cdkf_emission_moments_kernel" : "cdkf_ukf_tangent_kernel";
  if (!ut_model_sources(mdl, o, f_src, h_src, nth, &why, ekf)) {
    if (variant == 2) {
      set_error("emission moments of a custom emission: the model must be one the literal recursions take -- %s (drift_kind=%d state_dim=%d "
                "emission_dim=%d emission_kind=%d)", why.c_str(), mdl->drift_kind, mdl->state_dim, mdl->emission_dim, mdl->emission_kind);
      return CDKF_EUNSUPPORTED;
    }
    set_error("%s_loglik_grad: the tangent sweep of the literal %s recursion needs %s (drift_kind=%d state_dim=%d emission_dim=%d "
              "emission_kind=%d solver=%d state_order=%d num_iter=%d)", ekf ? "ekf" : "ukf", ekf ? "extended" : "unscented", why.c_str(), mdl->drift_kind,
              mdl->state_dim, mdl->emission_dim, mdl->emission_kind, o->solver, o->state_order, o->num_iter);
    return CDKF_EUNSUPPORTED;
  }
  const std::string src = generate_ut_source(mdl, f_src, h_src, bytes, variant);
  std::string arch = arch_override ? arch_override : "";
  int dev = 0;
  if (!arch_override) {
    CDKF_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CDKF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    arch = prop.gcnArchName;
  }
  // (-O1, as the workgroup variants: a loop-heavy kernel whose state lives in scratch by design; nothing to gain from unrolling it)
  const char* olevel = "-O1";
  const std::string tag = variant == 2 ? "emission moments" : ekf ? "ekf tangent" : "ukf tangent";
  const std::string cache_key = rtc_cache_key(src, arch, olevel, entry, tag);
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!arch_override) {
    auto it = g_ut_modules.find({dev, cache_key});
    if (it != g_ut_modules.end()) {
      *fn = it->second.fn;
      return CDKF_OK;
    }
  }
  std::vector<char> code;
  std::string unused;
  if (getenv("CDKF_CUSTOM_DUMP") || !rtc_cache_load(cache_key, code, unused)) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "cdkf_ukf_tangent.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      set_error("ukf tangent sweep: hiprtcCreateProgram failed");
      return CDKF_EHIP;
    }
    const std::string inc = "-I" + source_dir(), off = "--offload-arch=" + arch;
    const std::vector<std::string> extra = rtc_extra_options(tag);
    std::vector<const char*> opts = {off.c_str(), olevel, "-std=c++17", inc.c_str(), "-Wno-pass-failed"};
    for (const std::string& x : extra) opts.push_back(x.c_str());
    const hiprtcResult res = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (res != HIPRTC_SUCCESS) {
      size_t n = 0;
      hiprtcGetProgramLogSize(prog, &n);
      std::string log(n ? n : 1, '\0');
      if (n) hiprtcGetProgramLog(prog, &log[0]);
      std::string brief;
      size_t pos = 0;
      while (pos < log.size()) {
        size_t eol = log.find('\n', pos);
        if (eol == std::string::npos) eol = log.size();
        if (log.compare(pos, 21, "In file included from") != 0) brief.append(log, pos, eol - pos + 1);
        pos = eol + 1;
      }
      set_error("ukf tangent sweep: compilation failed (%s): %.600s", hiprtcGetErrorString(res), brief.c_str());
      hiprtcDestroyProgram(&prog);
      return CDKF_EINVAL;
    }
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    code.resize(sz);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    rtc_cache_store(cache_key, code, std::string(), tag + " d=" + std::to_string(mdl->state_dim) + " m=" + std::to_string(mdl->emission_dim) + " " + olevel);
    if (const char* dir = getenv("CDKF_CUSTOM_DUMP")) {
      const std::string base = std::string(dir) + (variant == 2 ? "/cdkf_emission_moments_" : ekf ? "/cdkf_ekf_tangent_" : "/cdkf_ukf_tangent_") + std::to_string(bytes) + "_" + cache_key.substr(0, 8);
      if (FILE* f = fopen((base + ".hip").c_str(), "w")) {
        fwrite(src.data(), 1, src.size(), f);
        fclose(f);
      }
      if (FILE* f = fopen((base + ".co").c_str(), "wb")) {
        fwrite(code.data(), 1, code.size(), f);
        fclose(f);
      }
    }
  }
  if (arch_override) return CDKF_OK;  // (compile check: no GPU)
  Compiled m;
  CDKF_HIP_CHECK(hipModuleLoadData(&m.module, code.data()));
  CDKF_HIP_CHECK(hipModuleGetFunction(&m.fn, m.module, entry));
  g_ut_modules[{dev, cache_key}] = m;
  *fn = m.fn;
  return CDKF_OK;
}

// compile check without a GPU (cdkf_ukf_tangent_compile): the kernel the model would get, built for gfx950
int ukf_tangent_compile_check(const cdkf_model* mdl, const cdkf_opts* o, int bytes_per_real, int ekf) {
  if (!mdl || !o || (bytes_per_real != 4 && bytes_per_real != 8)) return CDKF_EINVAL;
  hipFunction_t fn;
  return ut_get_function(mdl, o, bytes_per_real, &fn, "gfx950", ekf);  // (ekf: 0 unscented, 1 extended, 2 the emission-moments kernel)
}

// emission moments under an emission given as source (cdkf_custom_emission_moments_*): device pointers, rows = number of (m, P) pairs;
// covs / out_cov null: point estimates.  Declared where it is called (cdkf_api.hip), not in cdkf_launch.h.
template <typename R>
int launch_custom_emission_moments(const cdkf_model* mdl, const cdkf_opts* o, int ukf, int64_t rows, const R* t, const R* u, const R* means,
                                   const R* covs, R* out_mean, R* out_cov, hipStream_t stream) {
  if (!mdl || !o || rows < 0 || (rows > 0 && (!means || !out_mean))) {
    set_error("custom_emission_moments: bad arguments");
    return CDKF_EINVAL;
  }
  if (!mdl->emission_kind) {
    set_error("custom_emission_moments: the model's emission is linear (cdkf_emission_moments_* serves it)");
    return CDKF_EINVAL;
  }
  cdkf_opts oo = *o;  // (the integrator's settings do not matter here: ask for the model alone)
  oo.solver = CDKF_SOLVER_DOPRI5;
  oo.adaptive = 0;
  oo.forecast = 0;
  hipFunction_t fn;
  int rc = ut_get_function(mdl, &oo, (int)sizeof(R), &fn, nullptr, 2);
  if (rc || rows == 0) return rc;
  const int d = mdl->state_dim, m = mdl->emission_dim;
  std::vector<R> par;
  for (int k = 0; k < m * d; ++k) par.push_back(R(mdl->H[k]));
  for (int k = 0; k < m; ++k) par.push_back(R(mdl->h_bias[k]));
  for (int k = 0; k < m * m; ++k) par.push_back(R(mdl->R[k]));
  const size_t bytes = par.size() * sizeof(R);
  ParamLease lease(stream);
  rc = param_pool_acquire(bytes, &lease.slot);
  if (rc) return rc;
  std::memcpy(lease.slot->host, par.data(), bytes);
  CDKF_HIP_CHECK(hipMemcpyAsync(lease.slot->dev, lease.slot->host, bytes, hipMemcpyHostToDevice, stream));
  struct EmArgs {  // (= the struct of kEmissionMomentsKernel)
    const R* par; const R* t; const R* u; const R* mu; const R* P; R* ym; R* yc; long rows; R c, wm0, wc0, wi; int ukf;
  } a{};
  a.par = (const R*)lease.slot->dev;
  a.t = t; a.u = mdl->input_dim > 0 ? u : nullptr; a.mu = means; a.P = (covs && out_cov) ? covs : nullptr; a.ym = out_mean;
  a.yc = (covs && out_cov) ? out_cov : nullptr; a.rows = rows; a.ukf = ukf ? 1 : 0;
  const R alpha = R(o->ukf_alpha), n = R(d), lamb = alpha * alpha * (n + R(o->ukf_kappa)) - n;
  a.c = std::sqrt(n + lamb);
  a.wm0 = lamb / (n + lamb);
  a.wc0 = lamb / (n + lamb) + (R(1) - alpha * alpha + R(o->ukf_beta));
  a.wi = R(1) / (R(2) * (n + lamb));
  void* params[] = {(void*)&a};
  note_kernel("emission_moments_kernel<%s> (custom emission, d=%d m=%d, %s)", real_name<R>(), d, m, ukf ? "unscented" : "extended");
  CDKF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)((rows + 63) / 64), 1, 1, 64, 1, 1, 0, stream, params, nullptr));
  return lease.release();
}
template int launch_custom_emission_moments<float>(const cdkf_model*, const cdkf_opts*, int, int64_t, const float*, const float*, const float*,
                                                   const float*, float*, float*, hipStream_t);
template int launch_custom_emission_moments<double>(const cdkf_model*, const cdkf_opts*, int, int64_t, const double*, const double*,
                                                    const double*, const double*, double*, double*, hipStream_t);

// the argument struct and the parameter block (host copies) of the tangent sweep
template <typename R>
static void ut_fill(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, UtArgs<R>& a, std::vector<R>& par) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  par.clear();
  for (long k = 0; k < mdl->n_theta; ++k) par.push_back(R(mdl->theta[k]));
  for (int k = 0; k < d; ++k) par.push_back(R(mdl->m0[k]));
  for (int k = 0; k < d * d; ++k) par.push_back(R(mdl->P0[k]));
  {
    std::vector<R> packed(d * (d + 1) / 2);
    lql_packed<R>(mdl->L, mdl->Qc, d, 1.0, packed.data());  // (pairs i <= j, row-major)
    std::vector<R> full((size_t)d * d);
    int e = 0;
    for (int i = 0; i < d; ++i)
      for (int j = i; j < d; ++j, ++e) full[i * d + j] = full[j * d + i] = packed[e];
    par.insert(par.end(), full.begin(), full.end());
  }
  for (int k = 0; k < m * d; ++k) par.push_back(R(mdl->H[k]));
  for (int k = 0; k < m; ++k) par.push_back(R(mdl->h_bias[k]));
  for (int k = 0; k < m * m; ++k) par.push_back(R(mdl->R[k]));
  const R alpha = R(o->ukf_alpha), n = R(d);
  const R lamb = alpha * alpha * (n + R(o->ukf_kappa)) - n;
  a.dt0 = R(o->dt0);
  a.dt_final = R(o->dt_final);
  a.c = std::sqrt(n + lamb);
  a.wm0 = lamb / (n + lamb);
  a.wc0 = lamb / (n + lamb) + (R(1) - alpha * alpha + R(o->ukf_beta));
  a.wi = R(1) / (R(2) * (n + lamb));
  a.N = N;
  a.T = T;
  a.max_steps = (long)o->max_steps;
  a.num_iter = o->num_iter;
  a.order = o->state_order == CDKF_ORDER_SECOND ? 2 : 1;
  const SweepStrides ss = sweep_strides(o, N, T, d, m, false);
  a.t_sn = ss.t_sn; a.t_sk = ss.t_sk; a.y_sn = ss.y_sn; a.y_sk = ss.y_sk; a.y_si = ss.y_si;
  const int lin = (o->layout_in == CDKF_LAYOUT_SAME) ? o->layout : o->layout_in;
  const ArrayStrides us = layout_strides(lin, N, T, mdl->input_dim > 0 ? mdl->input_dim : 1);
  a.u_sn = us.sn; a.u_sk = us.sk; a.u_si = us.si;
  a.m_sn = ss.m_sn; a.m_sk = ss.m_sk; a.m_si = ss.m_si; a.P_sn = ss.P_sn; a.P_sk = ss.P_sk; a.P_si = ss.P_si;
}

template <typename R>
static int launch_tangent_impl(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                               R* grad_model, int32_t* status, hipStream_t stream, bool ekf, R* const* moments = nullptr) {
  hipFunction_t fn;
  int rc = ut_get_function(mdl, o, (int)sizeof(R), &fn, nullptr, ekf ? 1 : 0);
  if (rc) return rc;
  if (!t || !y || !ll || (!moments && !grad && mdl->n_theta > 0)) {
    set_error(moments ? "filter: t, y and ll must not be NULL" : "ukf_loglik_grad: t, y, ll and grad must not be NULL");
    return CDKF_EINVAL;
  }
  if (N < 1 || T < 1) return CDKF_OK;
  UtArgs<R> a{};
  std::vector<R> par;
  ut_fill<R>(mdl, o, N, T, a, par);
  if (moments) {  // value mode: a lane per trajectory, the filter's outputs
    a.value_only = 1;
    a.fm = moments[0]; a.fc = moments[1]; a.pm = moments[2]; a.pc = moments[3];
  }
  const size_t bytes = par.size() * sizeof(R);
  ParamLease lease(stream);
  rc = param_pool_acquire(bytes, &lease.slot);
  if (rc) return rc;
  std::memcpy(lease.slot->host, par.data(), bytes);
  CDKF_HIP_CHECK(hipMemcpyAsync(lease.slot->dev, lease.slot->host, bytes, hipMemcpyHostToDevice, stream));
  a.par = (const R*)lease.slot->dev;
  a.t = t; a.y = y; a.ll = ll; a.grad = grad; a.grad_model = grad_model; a.status = status;
  a.u = mdl->input_dim > 0 ? (const R*)o->inputs : nullptr;  // (device memory here: the host entry points have uploaded it)
  a.all = grad_model ? 1 : 0;
  const int d = mdl->state_dim, m = mdl->emission_dim, np = d * (d + 1) / 2, npm = m * (m + 1) / 2;
  const long nleaf = moments ? 1 : (a.all ? (long)mdl->n_theta + d + 2 * np + m * d + m + npm : (mdl->n_theta > 0 ? (long)mdl->n_theta : 1));
  const long total = N * nleaf;
  void* params[] = {(void*)&a};
  if (moments) note_kernel("%s_tangent_kernel<%s> (d=%d m=%d, value mode)", ekf ? "ekf" : "ukf", real_name<R>(), d, m);
  else note_kernel("%s_tangent_kernel<%s> (d=%d m=%d, %ld leaf entries)", ekf ? "ekf" : "ukf", real_name<R>(), d, m, nleaf);
  CDKF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)((total + 63) / 64), 1, 1, 64, 1, 1, 0, stream, params, nullptr));
  return lease.release();
}
// the FILTER through the same kernels (value mode): what launch_custom sends here -- an emission given as source above six dimensions
template <typename R>
static int launch_tangent_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm, R* fc,
                                 R* pm, R* pc, int32_t* status, hipStream_t stream, bool ekf) {
  cdkf_opts oo = *o;
  if (!ekf) {  // (the unscented filter has no state_order / num_iter: the tangent sweep's checks of them do not apply)
    oo.state_order = CDKF_ORDER_FIRST;
    oo.num_iter = 1;
  }
  R* const moments[4] = {fm, fc, pm, pc};
  return launch_tangent_impl<R>(mdl, &oo, N, T, t, y, ll, nullptr, nullptr, status, stream, ekf, moments);
}

template <typename R>
int launch_ukf_tangent(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                       R* grad_model, int32_t* status, hipStream_t stream) {
  return launch_tangent_impl<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, false);
}
template <typename R>
int launch_ekf_tangent(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                       R* grad_model, int32_t* status, hipStream_t stream) {
  return launch_tangent_impl<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, true);
}
template int launch_ekf_tangent<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, float*, float*, float*,
                                       int32_t*, hipStream_t);
template int launch_ekf_tangent<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*, double*, double*,
                                        double*, int32_t*, hipStream_t);
template int launch_ukf_tangent<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, float*, float*, float*,
                                       int32_t*, hipStream_t);
template int launch_ukf_tangent<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*, double*, double*,
                                        double*, int32_t*, hipStream_t);

// host copies of the sweep's argument struct and parameter block (cdkf_debug_ukf_tangent_args: the CPU-sanitizer build of the kernel)
int ukf_tangent_debug_args(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int bytes_per_real, int all, void* args_out,
                           int64_t args_cap, void* par_out, int64_t par_cap) {
  if (!mdl || !o || !args_out || !par_out || (bytes_per_real != 4 && bytes_per_real != 8)) return CDKF_EINVAL;
  if (bytes_per_real == 8) {
    UtArgs<double> a{};
    std::vector<double> par;
    ut_fill<double>(mdl, o, N, T, a, par);
    a.all = all == 1;
  a.value_only = all == 2;
    if ((int64_t)sizeof(a) > args_cap || (int64_t)(par.size() * 8) > par_cap) return CDKF_EINVAL;
    std::memcpy(args_out, &a, sizeof(a));
    std::memcpy(par_out, par.data(), par.size() * 8);
    return (int)par.size();
  }
  UtArgs<float> a{};
  std::vector<float> par;
  ut_fill<float>(mdl, o, N, T, a, par);
  a.all = all == 1;
  a.value_only = all == 2;
  if ((int64_t)sizeof(a) > args_cap || (int64_t)(par.size() * 4) > par_cap) return CDKF_EINVAL;
  std::memcpy(args_out, &a, sizeof(a));
  std::memcpy(par_out, par.data(), par.size() * 4);
  return (int)par.size();
}

// the argument blocks of launch_custom for the CPU-sanitizer build of the generated kernel (cdkf_debug_custom_reg_blob): no HIP call
int custom_debug_reg_blob(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int algo, int bytes_per_real, void* par_out,
                          int64_t par_cap_bytes, int64_t* ip_out) {
  if (!mdl || !o || !par_out || !ip_out || (bytes_per_real != 4 && bytes_per_real != 8) || algo < 0 || algo > 3) {
    set_error("cdkf_debug_custom_reg_blob: bad arguments");
    return CDKF_EINVAL;
  }
  long ip[26];
  int64_t n = 0;
  unsigned blocks = 0;
  if (bytes_per_real == 8) {
    std::vector<double> par;
    blocks = custom_reg_blob<double>(mdl, o, N, T, algo == 2, algo == 3, false, par, ip).blocks;
    n = (int64_t)par.size();
    if (n * 8 > par_cap_bytes) return CDKF_EINVAL;
    std::memcpy(par_out, par.data(), n * 8);
  } else {
    std::vector<float> par;
    blocks = custom_reg_blob<float>(mdl, o, N, T, algo == 2, algo == 3, false, par, ip).blocks;
    n = (int64_t)par.size();
    if (n * 4 > par_cap_bytes) return CDKF_EINVAL;
    std::memcpy(par_out, par.data(), n * 4);
  }
  for (int k = 0; k < 26; ++k) ip_out[k] = ip[k];
  ip_out[26] = blocks;
  return (int)n;
}

void custom_rtc_cache_stats(int64_t* hits, int64_t* misses) {
  if (hits) *hits = g_rtc_hits.load();
  if (misses) *misses = g_rtc_misses.load();
}

void custom_set_source_dir(const char* dir) {
  std::lock_guard<std::mutex> lock(g_mutex_srcdir());
  g_src_dir = dir ? dir : "";
}

}  // namespace cdkf
