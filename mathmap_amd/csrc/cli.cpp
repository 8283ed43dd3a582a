// placeholder CLI
#include <cstdio>
int main() { printf("mathmap_hip_cli\n"); return 0; }
