// Command-line harness around dot_ring_amd/csrc/hostsigma.hpp (the small-call host route of the Tiny / Thin / Pedersen VRFs), plain g++,
// no GPU: tests/test_hostsigma_cpu.py feeds it the reference's own test vectors.  One command per line on stdin, fields in hex
// ("-" = empty):
//   suite <suite_id> <xof 0|1> <generator xy> <blinding base xy>
//   prove <tiny|thin|pedersen> <alpha> <ad> <salt> <secret>                -> "proof <hex> aux <hex>" | "error <code>"
//   verify <tiny|thin> <proof> <public key> <input> <ad> <salt>            -> "verdict <0..3>"
//   verify pedersen <proof> - <input> <ad> <salt>                          -> "verdict <0..3>"
//   decode <32-byte encoding>                                              -> "point <xy hex>" | "invalid"
//   mul <secret|public|fixed> <point xy> <scalar le32>                     -> "point <xy hex>"
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "hostsigma.hpp"

using namespace drh;

static Bytes unhex(const std::string& s) {
    Bytes out;
    if (s == "-") return out;
    for (size_t i = 0; i + 1 < s.size(); i += 2) out.push_back((uint8_t)std::stoul(s.substr(i, 2), nullptr, 16));
    return out;
}
static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}
static Span sp(const Bytes& b) { return Span{b.data(), b.size()}; }

int main() {
    VrfSuite su;
    std::shared_ptr<const TeSuiteTables> tb;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string cmd;
        in >> cmd;
        if (cmd == "suite") {
            std::string id, xof, g, b;
            in >> id >> xof >> g >> b;
            su.suite_id = unhex(id);
            su.xof = xof == "1";
            Bytes gb = unhex(g), bb = unhex(b);
            if (gb.size() != 64 || bb.size() != 64) { std::puts("error bad suite"); continue; }
            std::memcpy(su.generator, gb.data(), 64);
            std::memcpy(su.blinding_base, bb.data(), 64);
            su.cv = te_curve(0);
            tb = te_suite_tables(su);
            std::puts(tb ? "ok" : "error tables");
        } else if (cmd == "prove") {
            std::string scheme, a, ad, salt, sk;
            in >> scheme >> a >> ad >> salt >> sk;
            Bytes ab = unhex(a), adb = unhex(ad), sb = unhex(salt), skb = unhex(sk);
            uint8_t out[192], aux[288];
            int rc;
            size_t plen, alen;
            if (scheme == "pedersen") { rc = pedersen_prove_one(su, *tb, sp(ab), sp(adb), sp(sb), skb.data(), out, aux); plen = 192; alen = 288; }
            else { const bool thin = scheme == "thin"; rc = ietf_prove_one(su, *tb, thin, sp(ab), sp(adb), sp(sb), skb.data(), out, aux); plen = thin ? 96 : 80; alen = 128; }
            if (rc) std::printf("error %d\n", rc);
            else std::printf("proof %s aux %s\n", hex(out, plen).c_str(), hex(aux, alen).c_str());
        } else if (cmd == "verify") {
            std::string scheme, pr, pk, inp, ad, salt;
            in >> scheme >> pr >> pk >> inp >> ad >> salt;
            Bytes prb = unhex(pr), pkb = unhex(pk), ib = unhex(inp), adb = unhex(ad), sb = unhex(salt);
            int v;
            if (scheme == "pedersen") v = prb.size() == 192 ? pedersen_verify_one(su, *tb, prb.data(), sp(ib), sp(adb), sp(sb)) : 3;
            else {
                const bool thin = scheme == "thin";
                v = (prb.size() == (thin ? 96u : 80u) && pkb.size() == 32) ? ietf_verify_one(su, *tb, thin, prb.data(), pkb.data(), sp(ib), sp(adb), sp(sb)) : 3;
            }
            std::printf("verdict %d\n", v);
        } else if (cmd == "decode") {
            std::string e;
            in >> e;
            Bytes eb = unhex(e);
            uint8_t xy[64];
            if (eb.size() == 32 && te_decode_checked(*te_curve(0), eb.data(), xy)) std::printf("point %s\n", hex(xy, 64).c_str());
            else std::puts("invalid");
        } else if (cmd == "mul") {
            std::string how, p, k;
            in >> how >> p >> k;
            Bytes pb = unhex(p), kb = unhex(k);
            TeExt P;
            uint64_t kk[4];
            uint8_t xy[64];
            if (pb.size() != 64 || kb.size() != 32 || !te_load_affine(pb.data(), P)) { std::puts("invalid"); continue; }
            load_le32(kb.data(), kk);
            const TeHostParams c = te_host_params(*te_curve(0));
            TeExt r;
            if (how == "secret") r = te_mul_secret(P, kk, c);
            else if (how == "public") { const uint64_t ks[1][4] = {{kk[0], kk[1], kk[2], kk[3]}}; r = te_msm_public(&P, ks, 1, c); }
            else { const TeFixedTable ft = te_fixed_table(P, c); r = te_add(te_mul_fixed(ft, kk, c, true), te_neg(te_mul_fixed(ft, kk, c, false)), c);
                   r = te_add(r, te_mul_fixed(ft, kk, c, true), c); }      // secret - public + secret picks: both must agree
            te_store_affine(r, xy);
            std::printf("point %s\n", hex(xy, 64).c_str());
        } else if (!cmd.empty()) {
            std::puts("error unknown command");
        }
        std::fflush(stdout);
    }
    return 0;
}
