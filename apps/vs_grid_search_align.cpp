// vs_grid_search_align -- search the aligner parameters that minimise output jitter (the role of the reference's
// grid_search_align.cpp:62-219).   vs_grid_search_align clip.y4m [-j N] [--device D | --devices a,b,...|all] [--frames M]
// Grid: phase_correlate x threshold x smallest_fraction x max_displacement, stabilizer with the smoother off, lag 1,
// smoother_memory 0 (grid_search_align.cpp:135-167).
#include <sstream>
#include "grid_runner.hpp"

int main(int argc, char** argv) {
    vsh::GridArgs args;
    if (!args.parse(argc, argv)) { std::cerr << "Usage: " << argv[0] << " video.y4m [-j N] [--device D | --devices a,b,...|all] [--frames M]" << std::endl; return 1; }
    try {
        std::vector<std::unique_ptr<vsh::DeviceClip>> clip;      // one copy of the clip per device slot
        double input_jitter = 0.0;
        if (!vsh::prepare(args, clip, input_jitter)) return 1;

        const bool phase_vals[] = {false, true};
        const double thresh_vals[] = {0.02, 0.03, 0.05};
        const float frac_vals[] = {0.3f, 0.5f, 0.8f};
        const double max_disp_vals[] = {6.0, 8.0, 10.0};
        std::vector<vsh::GridCombo> combos;
        for (bool pc : phase_vals)
            for (double thr : thresh_vals)
                for (float frac : frac_vals)
                    for (double md : max_disp_vals) {
                        vsh::GridCombo c;
                        vs_stabilizer_params_default(&c.params);
                        c.params.enable_smoother = 0;
                        c.params.lag = 1;
                        c.params.smoother_memory = 0;
                        c.params.aligner.phase_correlate = pc ? 1 : 0;
                        c.params.aligner.threshold = thr;
                        c.params.aligner.smallest_fraction = frac;
                        c.params.aligner.max_displacement = md;
                        std::ostringstream s;
                        s << "PC=" << pc << " thr=" << thr << " frac=" << frac << " maxDisp=" << md;
                        c.label = s.str();
                        combos.push_back(c);
                    }
        std::cerr << vsjit::score_note() << std::endl;
        std::cout << "Running " << combos.size() << " parameter combinations using " << args.jobs << " threads" << std::endl;
        const vsh::GridResult r = vsh::run_grid(clip, input_jitter, combos, args.jobs);
        if (args.dump) vsh::dump_ratios(r);
        if (r.best < 0) { std::cerr << "No combination produced output." << std::endl; return 1; }
        const vs_aligner_params& b = combos[(size_t)r.best].params.aligner;
        std::cout << "\nBest params: phase_correlate=" << b.phase_correlate << "  threshold=" << b.threshold
                  << "  smallest_fraction=" << b.smallest_fraction << "  max_displacement=" << b.max_displacement
                  << "  ratio=" << r.best_ratio << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
