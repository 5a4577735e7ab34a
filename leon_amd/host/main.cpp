// main.cpp -- same shape as /root/reference/src/main.cpp:37-51: version banner, Leon().run(argc, argv) in a try block,
// "EXCEPTION: <msg>" on stderr and EXIT_FAILURE when the tool throws.
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "leon_host.hpp"

static void displayVersion(std::ostream& os) {
    os << "* * * * * * * * * * * * * * * * * * * * * *" << std::endl;
    os << "* leon_amd DNA encode path, C-ABI version " << leon_dna_abi_version() << "  *" << std::endl;
    os << "* * * * * * * * * * * * * * * * * * * * * *" << std::endl;
}

int main(int argc, char* argv[]) {
    if (argc > 1 && (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-v") == 0)) {
        displayVersion(std::cout);
        return EXIT_FAILURE;                  // the reference returns EXIT_FAILURE after the banner (main.cpp:40)
    }
    try {
        leon_host::Leon().run(argc, argv);
    } catch (leon_host::Exception& e) {
        std::cerr << "EXCEPTION: " << e.getMessage() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
