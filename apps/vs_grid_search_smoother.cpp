// vs_grid_search_smoother -- search the smoother / decay parameters that minimise output jitter (the role of the
// reference's grid_search_smoother.cpp:91-287).   vs_grid_search_smoother clip.y4m [-j N] [--device D | --devices a,b,...|all] [--frames M] [--quick]
// Grid: lag x smoother_memory x lambda x (min_disp < max_disp) x (min_decay > max_decay), aligner defaults
// (grid_search_smoother.cpp:160-186); --quick keeps the first value of the displacement / decay axes (48 combinations).
#include <sstream>
#include "grid_runner.hpp"

int main(int argc, char** argv) {
    bool quick = false;
    std::vector<char*> rest;
    for (int i = 0; i < argc; i++) {
        if (std::string(argv[i]) == "--quick") quick = true;
        else rest.push_back(argv[i]);
    }
    vsh::GridArgs args;
    if (!args.parse((int)rest.size(), rest.data())) { std::cerr << "Usage: " << argv[0] << " video.y4m [-j N] [--device D | --devices a,b,...|all] [--frames M] [--quick]" << std::endl; return 1; }
    try {
        std::vector<std::unique_ptr<vsh::DeviceClip>> clip;      // one copy of the clip per device slot
        double input_jitter = 0.0;
        if (!vsh::prepare(args, clip, input_jitter)) return 1;

        const std::vector<int> lag_vals = {3, 5, 8, 10};
        const std::vector<int> mem_vals = {5, 8, 10};
        const std::vector<double> lambda_vals = {4.0, 6.0, 8.0, 10.0};
        std::vector<double> min_disp_vals = {16.0, 32.0, 48.0}, max_disp_vals = {64.0, 96.0, 128.0};
        std::vector<double> min_decay_vals = {0.99, 0.95, 0.9}, max_decay_vals = {0.7, 0.5, 0.3};
        if (quick) { min_disp_vals.resize(1); max_disp_vals.resize(1); min_decay_vals.resize(1); max_decay_vals.resize(1); }
        std::vector<vsh::GridCombo> combos;
        for (int lag : lag_vals)
            for (int mem : mem_vals)
                for (double lam : lambda_vals)
                    for (double mind : min_disp_vals)
                        for (double maxd : max_disp_vals) {
                            if (!(mind < maxd)) continue;
                            for (double mindec : min_decay_vals)
                                for (double maxdec : max_decay_vals) {
                                    if (!(mindec > maxdec)) continue;
                                    vsh::GridCombo c;
                                    vs_stabilizer_params_default(&c.params);
                                    c.params.lag = lag;
                                    c.params.smoother_memory = mem;
                                    c.params.lambda = lam;
                                    c.params.min_disp = mind; c.params.max_disp = maxd;
                                    c.params.min_decay = mindec; c.params.max_decay = maxdec;
                                    std::ostringstream s;
                                    s << "lag=" << lag << " mem=" << mem << " lambda=" << lam << "  minDisp=" << mind << " maxDisp=" << maxd
                                      << "  minDecay=" << mindec << " maxDecay=" << maxdec;
                                    c.label = s.str();
                                    combos.push_back(c);
                                }
                        }
        std::cerr << vsjit::score_note() << std::endl;
        std::cout << "Evaluating " << combos.size() << " parameter combinations using " << args.jobs << " threads" << std::endl;
        const vsh::GridResult r = vsh::run_grid(clip, input_jitter, combos, args.jobs);
        if (args.dump) vsh::dump_ratios(r);
        if (r.best < 0) { std::cerr << "No combination produced output." << std::endl; return 1; }
        const vs_stabilizer_params& b = combos[(size_t)r.best].params;
        std::cout << "\nBest parameters:" << std::endl;
        std::cout << "  lag             = " << b.lag << std::endl;
        std::cout << "  smoother_memory = " << b.smoother_memory << std::endl;
        std::cout << "  lambda          = " << b.lambda << std::endl;
        std::cout << "  min_disp        = " << b.min_disp << std::endl;
        std::cout << "  max_disp        = " << b.max_disp << std::endl;
        std::cout << "  min_decay       = " << b.min_decay << std::endl;
        std::cout << "  max_decay       = " << b.max_decay << std::endl;
        std::cout << "  jitter ratio    = " << r.best_ratio << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
