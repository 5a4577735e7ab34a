// Command-line front of the host mirror.  Behaviour kept from the reference's entry point (/root/reference/src/main.cpp:37-51):
//   * `-v` / `--version` print a banner and the process ends with EXIT_FAILURE (yes, failure: that is what upstream returns);
//   * anything the tool throws is reported as "EXCEPTION: <message>" on stderr, exit status EXIT_FAILURE;
//   * otherwise the tool parses its own arguments (Leon::run) and the process ends with EXIT_SUCCESS.
#include <cstdlib>
#include <iostream>
#include <string>

#include "leon_host.hpp"

namespace {

bool wants_version(int argc, char** argv) {
    if (argc < 2) return false;
    const std::string first(argv[1]);
    return first == "-v" || first == "--version";
}

int report(const leon_host::Exception& failure) {
    std::cerr << "EXCEPTION: " << failure.getMessage() << std::endl;
    return EXIT_FAILURE;
}

}  // namespace

int main(int argc, char* argv[]) {
    if (wants_version(argc, argv)) {
        const std::string rule(44, '*');
        std::cout << rule << "\n* leon_amd DNA encode path, C-ABI version " << leon_dna_abi_version()
                  << "\n* .leon streams follow a restatement of gatb-core's\n* coders: byte parity with reference Leon is unverified\n" << rule << std::endl;
        return EXIT_FAILURE;
    }
    leon_host::Leon tool;
    try {
        // host logic that needs no GPU, for the CPU test-suite: the .leon container layer and the FASTA / FASTQ reader
        if (argc == 3 && std::string(argv[1]) == "-selftest-container") return leon_host::selftest_container(argv[2]);
        if (argc == 3 && std::string(argv[1]) == "-selftest-bank") return leon_host::selftest_bank(argv[2]);
        tool.run(argc, argv);
    } catch (const leon_host::Exception& failure) {
        return report(failure);
    }
    return EXIT_SUCCESS;
}
