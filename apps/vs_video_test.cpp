// vs_video_test -- stabilize every clip of a directory (the role of the reference's video_test.cpp:10-128).
//   vs_video_test [input_dir=../recordings] [output_dir=output] [--chunk N] [--device D] [--crop N] [--bilinear | --lanczos2] [--444]
// For each .y4m / .bgr clip writes output_dir/processed_<name>.  Like the reference's driver it runs the default
// VideoStabilizerParams with crop_pixels = 0 (video_test.cpp:54-55) unless --crop is given.  Frames go to the GPU in
// chunks of N (default 64) and through vs_stabilizer_process_batch, which is defined as N successive processFrame calls.
#include <chrono>
#include <filesystem>
#include <iostream>
#include "harness.hpp"

namespace fs = std::filesystem;

static bool process_clip(const std::string& in_path, const std::string& out_path, vs_stabilizer_params params, int device, int chunk,
                         bool force444) {
    vsio::Reader reader;
    if (!reader.open(in_path)) { std::cerr << "Error: " << reader.error << std::endl; return false; }
    const vsio::Format fin = reader.fmt;
    const int crop = params.crop_pixels > 0 ? params.crop_pixels : 0;
    vsio::Format fout = fin;
    fout.w = fin.w - 2 * crop; fout.h = fin.h - 2 * crop;             // video_test.cpp:74-76
    if (force444 || fin.chroma == vsio::Chroma::Mono) fout.chroma = vsio::Chroma::C444;
    if (fout.w <= 0 || fout.h <= 0) { std::cerr << "Error: crop larger than the frame" << std::endl; return false; }
    vsio::Writer writer;
    if (!writer.open(out_path, fout)) { std::cerr << "Error: " << writer.error << std::endl; return false; }
    std::cout << "Input FPS: " << (double)fin.fps_num / fin.fps_den << " | Frame Size: " << fout.w << "x" << fout.h << std::endl;
    std::cout << "Writing processed video to: " << out_path << std::endl;

    const size_t esz = fin.bits > 8 ? 2 : 1;
    const size_t in_elems = fin.bgr_elems(), out_elems = fout.bgr_elems();
    std::vector<uint8_t> h_in((size_t)chunk * in_elems * esz), h_out((size_t)chunk * out_elems * esz);
    vsh::DeviceBuffer d_in(h_in.size()), d_out(h_out.size());
    std::vector<int32_t> has_output((size_t)chunk);
    vs_stabilizer* stab = vs_stabilizer_create(&params, device);
    if (!stab) { std::cerr << "Error: vs_stabilizer_create: " << vs_last_error() << std::endl; return false; }

    long frame_count = 0, written = 0, next_report = 100;
    const auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (;;) {
        int n = 0;
        while (n < chunk && reader.next(h_in.data() + (size_t)n * in_elems * esz)) n++;
        if (n == 0) break;
        vsh::hip_check(hipMemcpy(d_in.ptr, h_in.data(), (size_t)n * in_elems * esz, hipMemcpyHostToDevice), "hipMemcpy H2D");
        int ow = 0, oh = 0;
        const int r = vs_stabilizer_process_batch(stab, d_in.ptr, in_elems, n, fin.w, fin.h, fin.w * 3, vsh::vs_format_of(fin), VS_MEM_DEVICE,
                                                  d_out.ptr, out_elems, has_output.data(), &ow, &oh);
        if (r < 0) { std::cerr << "Error: vs_stabilizer_process_batch: " << vs_last_error() << std::endl; ok = false; break; }
        if (r > 0) vsh::hip_check(hipMemcpy(h_out.data(), d_out.ptr, (size_t)n * out_elems * esz, hipMemcpyDeviceToHost), "hipMemcpy D2H");
        for (int i = 0; i < n; i++) {
            // the reference hands the empty Mat of the first `lag` frames to the writer, which drops it (video_test.cpp:104-109)
            if (has_output[i]) { writer.write(h_out.data() + (size_t)i * out_elems * esz); written++; }
        }
        frame_count += n;
        while (frame_count >= next_report) { std::cout << "Processed " << next_report << " frames..." << std::endl; next_report += 100; }
        if (n < chunk) break;
    }
    if (!reader.error.empty()) { std::cerr << "Error: " << reader.error << std::endl; ok = false; }
    vs_stabilizer_destroy(stab);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Finished processing " << frame_count << " frames (" << written << " written, " << frame_count / std::max(sec, 1e-9)
              << " frames/s incl. file I/O) for video: " << fs::path(in_path).filename().string() << std::endl;
    return ok;
}

int main(int argc, char** argv) {
    std::string input_dir = "../recordings", output_dir = "output";   // video_test.cpp:12-13
    int chunk = 64, device = 0, crop = 0, positional = 0;
    bool bilinear = false, lanczos2 = false, force444 = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--chunk" && i + 1 < argc) chunk = std::max(1, std::atoi(argv[++i]));
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--crop" && i + 1 < argc) crop = std::atoi(argv[++i]);
        else if (a == "--bilinear") bilinear = true;
        else if (a == "--lanczos2") lanczos2 = true;
        else if (a == "--444") force444 = true;
        else if (positional == 0) { input_dir = a; positional++; }
        else if (positional == 1) { output_dir = a; positional++; }
        else { std::cerr << "Usage: " << argv[0] << " [input_dir] [output_dir] [--chunk N] [--device D] [--crop N] [--bilinear] [--444]\n"; return EXIT_FAILURE; }
    }
    try {
        if (!fs::exists(output_dir)) {
            fs::create_directories(output_dir);
            std::cout << "Created output directory: " << output_dir << std::endl;
        }
        if (!fs::is_directory(input_dir)) { std::cerr << "Error: Input directory does not exist or is not a directory.\n"; return EXIT_FAILURE; }
        std::vector<std::string> clips;
        for (const auto& e : fs::directory_iterator(input_dir)) {
            const std::string ext = e.path().extension().string();
            if (e.is_regular_file() && (ext == ".y4m" || ext == ".bgr")) clips.push_back(e.path().filename().string());
        }
        std::sort(clips.begin(), clips.end());
        if (clips.empty()) { std::cerr << "No .y4m / .bgr clips found in the input directory: " << input_dir << std::endl; return EXIT_FAILURE; }
        if (vs_device_count() <= device) { std::cerr << "Error: no HIP device " << device << std::endl; return EXIT_FAILURE; }

        vs_stabilizer_params params;
        vs_stabilizer_params_default(&params);
        params.crop_pixels = crop;                         // 0: disable crop so we can see what it is doing (video_test.cpp:55)
        if (bilinear) params.warp_mode = VS_WARP_BILINEAR;    // the Halide sampler's float lerp (the default is cv::warpAffine's fixed-point bilinear, VS_WARP_BILINEAR_CV)
        if (lanczos2) params.warp_mode = VS_WARP_LANCZOS2;
        int failed = 0;
        for (const auto& name : clips) {
            const std::string in_path = (fs::path(input_dir) / name).string();
            std::cout << "\nProcessing video: " << in_path << std::endl;
            if (!process_clip(in_path, (fs::path(output_dir) / ("processed_" + name)).string(), params, device, chunk, force444)) failed++;
        }
        if (failed) { std::cerr << "\n" << failed << " clip(s) failed." << std::endl; return EXIT_FAILURE; }
        std::cout << "\nAll videos have been processed successfully." << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
