// Minimal command-line flags for the tools: `--key value`, `--key=value`, and bare `--key` for booleans -- the three
// forms the reference's simple_flags accepts (tools/simple_flags.h:18-120).  Unknown flags are reported.
#pragma once
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace tools {

class Args final {
public:
    void Def(const std::string& name, const std::string& dflt, const std::string& help, bool is_bool = false) {
        opts_[name] = Opt{dflt, help, is_bool};
        order_.push_back(name);
    }
    bool Parse(int argc, char** argv) {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i], val;
            bool has_val = false;
            const size_t eq = a.find('=');
            if (eq != std::string::npos) {
                val = a.substr(eq + 1);
                a = a.substr(0, eq);
                has_val = true;
            }
            auto it = opts_.find(a);
            if (it == opts_.end()) {
                unknown_.push_back(a);
                continue;
            }
            if (it->second.is_bool) {
                if (!has_val && i + 1 < argc && IsBoolWord(argv[i + 1])) { val = argv[++i]; has_val = true; }
                it->second.value = has_val ? val : "true";
            } else {
                if (!has_val) {
                    if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; return false; }
                    val = argv[++i];
                }
                it->second.value = val;
            }
        }
        if (!unknown_.empty()) {
            std::cerr << "unknown option(s):";
            for (auto& u : unknown_) std::cerr << " '" << u << "'";
            std::cerr << "\n";
            return false;
        }
        return true;
    }
    std::string Str(const std::string& n) const { return opts_.at(n).value; }
    int Int(const std::string& n) const { return std::atoi(opts_.at(n).value.c_str()); }
    long long I64(const std::string& n) const { return std::atoll(opts_.at(n).value.c_str()); }
    double Num(const std::string& n) const { return std::atof(opts_.at(n).value.c_str()); }
    bool Bool(const std::string& n) const {
        const std::string& v = opts_.at(n).value;
        return v == "true" || v == "1" || v == "on" || v == "yes";
    }
    void PrintHelp() const {
        for (auto& n : order_) std::cout << "  " << n << " (default: " << opts_.at(n).value << ")  " << opts_.at(n).help << "\n";
    }

private:
    static bool IsBoolWord(const std::string& s) { return s == "true" || s == "false" || s == "0" || s == "1" || s == "on" || s == "off"; }
    struct Opt {
        std::string value, help;
        bool is_bool;
    };
    std::map<std::string, Opt> opts_;
    std::vector<std::string> order_, unknown_;
};

}  // namespace tools
