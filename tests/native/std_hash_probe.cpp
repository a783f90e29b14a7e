// Prints std::hash<float> of the container's libstdc++ for the floats given as hex bit patterns on the command line, and the combine chain the reference's
// SetupWorkItems runs over them (libraries/omm-lib/src/bake_cpu_impl.cpp:626-631) with std::hash<glm::vec2> restated from glm/gtx/hash.inl (glm is not in
// the image).  tests/test_oracle_units.py compares the oracle's restatement with this output: the std::hash<float> part is the real library's answer.
//   g++ -O1 -o std_hash_probe std_hash_probe.cpp && ./std_hash_probe 3e800000 3f400000 ...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        const uint32_t b = (uint32_t)strtoul(argv[i], nullptr, 16);
        float f; memcpy(&f, &b, 4);
        printf("%08x %016llx\n", b, (unsigned long long)std::hash<float>()(f));
    }
    printf("int %016llx %016llx\n", (unsigned long long)std::hash<int32_t>()(-3), (unsigned long long)std::hash<int32_t>()(7));
    return 0;
}
