// Compile-time (and, with a GPU, run-time) check of include/pigo.hpp against libpigo_hip.so.
// Built by tests/test_abi_cpu.py::test_cpp_mirror_compiles_and_links; with no GPU it only verifies that the
// reference-shaped API compiles, links and reports the missing device as an exception instead of aborting.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "pigo.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary), g(argv[2], std::ios::binary);
    std::vector<uint8_t> packet((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> gray((std::istreambuf_iterator<char>(g)), std::istreambuf_iterator<char>());
    try {
        pigo::Pigo pg = pigo::NewPigo().Unpack(packet);
        pigo::CascadeParams cp{{&gray, 400, 320, 320}, 20, 1000, 0.2, 1.1};  // core/pigo_test.go:44-50
        std::vector<pigo::Detection> dets = pg.RunCascade(cp, 0.0);
        std::vector<pigo::Detection> cl = pg.ClusterDetections(dets, 0.1);
        std::printf("dets=%zu clusters=%zu", dets.size(), cl.size());
        for (const auto &c : cl) std::printf(" (%d,%d,%d,%.4f)", c.Row, c.Col, c.Scale, c.Q);
        // core/grayscale_test.go:14-34: a uniform (177,177,177,255) image stays 177
        std::vector<uint8_t> pix(10 * 10 * 4, 177);
        for (size_t i = 3; i < pix.size(); i += 4) pix[i] = 255;
        const std::vector<uint8_t> g177 = pigo::RgbToGrayscale(pigo::Image{&pix, 40, 10, 10, PIGO_PIX_RGBA});
        bool gray_ok = g177.size() == 100;
        for (uint8_t v : g177) gray_ok = gray_ok && v == 177;
        std::printf(" gray177=%d", gray_ok ? 1 : 0);
        bool eye_ok = true;
        if (argc > 3) {  // core/puploc_test.go:34-80 on the cluster found above, with a fixed rand.Float32() stream
            std::ifstream pf(argv[3], std::ios::binary);
            std::vector<uint8_t> ppk((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
            pigo::PuplocCascade plc = pigo::NewPuplocCascade().UnpackCascade(ppk);
            pigo::DetectorState st;
            uint32_t lcg = 12345u;
            st.Float32 = [&lcg]() { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) / 16777216.0f; };
            const pigo::Detection &d = cl[0];
            pigo::Puploc req{d.Row - (int)(0.075f * (float)d.Scale), d.Col - (int)(0.175f * (float)d.Scale), (float)d.Scale * 0.25f, 63};
            const pigo::Puploc eye = plc.RunDetector(req, cp.ImageParams, 0.0, false, st);
            eye_ok = eye.Row > d.Row - d.Scale / 2 && eye.Row < d.Row && eye.Col > d.Col - d.Scale / 2 && eye.Col < d.Col;
            std::printf(" eye=(%d,%d,%.3f) eye_ok=%d", eye.Row, eye.Col, eye.Scale, eye_ok ? 1 : 0);
        }
        // batch / multi-GPU extension: a plan for the same parameters, a world-size-1 communicator (no RCCL needed), the shard
        // arithmetic and the wire codec (host side of what pigo_run_batch_sharded packs on the device)
        bool wire_ok = true;
        {
            const pigo::Plan plan(pg, 400, 320, 320, 20, 1000, 0.2, 1.1, 0.0, 8, 256);
            wire_ok = wire_ok && plan.Info().max_frames == 8 && plan.Info().windows_per_frame > 0;
            const pigo::Comm comm(nullptr, 0, 1, 0);
            wire_ok = wire_ok && comm.Rank() == 0 && comm.World() == 1;
            const auto b0 = pigo::ShardBounds(10, 0, 4), b3 = pigo::ShardBounds(10, 3, 4);
            wire_ok = wire_ok && b0.first == 0 && b0.second == 3 && b3.first == 8 && b3.second == 10;
            const int gcap = 4;
            std::vector<pigo_det> lists(2 * 8);
            const int32_t counts[2] = {2, 6};  // the second list is longer than gather_cap: truncated row, true count kept
            for (int i = 0; i < 16; ++i) lists[(size_t)i] = pigo_det{i, 2 * i, 3 * i, 0.5f * (float)i};
            std::vector<int32_t> wire(3 * pigo_wire_words(gcap));
            pigo::detail::check(pigo_pack_lists(lists.data(), counts, 2, 3, 8, gcap, wire.data()), "pigo_pack_lists");
            int tc = -1, fl = -1;
            const auto l0 = pigo::UnpackList(wire.data(), gcap, &tc, &fl);
            wire_ok = wire_ok && l0.size() == 2 && tc == 2 && fl == 0 && l0[1].Row == 1 && l0[1].Col == 2 && l0[1].Q == 0.5f;
            const auto l1 = pigo::UnpackList(wire.data() + pigo_wire_words(gcap), gcap, &tc, &fl);
            wire_ok = wire_ok && l1.size() == 4 && tc == 6 && fl == PIGO_WIRE_TRUNCATED_GATHER && l1[3].Row == 11;
            const auto l2 = pigo::UnpackList(wire.data() + 2 * pigo_wire_words(gcap), gcap, &tc, &fl);
            wire_ok = wire_ok && l2.empty() && tc == 0 && fl == PIGO_WIRE_PADDING;  // padding row
        }
        std::printf(" wire_ok=%d", wire_ok ? 1 : 0);
        std::printf("\n");
        return cl.empty() || !gray_ok || !eye_ok || !wire_ok ? 1 : 0;
    } catch (const pigo::Panic &e) {
        std::printf("panic: %s\n", e.what());
        return 3;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 4;
    }
}
