# Native build of the MI355X xPNG stack.  Everything lands IN-TREE (xpng_amd/lib, xpng_amd/bin) so the
# built objects travel to the GPU box with the repo snapshot.
HIPCC  ?= hipcc
CC     ?= gcc
ARCH   ?= gfx950
CSRC   := xpng_amd/csrc
LIB    := xpng_amd/lib
BIN    := xpng_amd/bin
HIPSRC := $(CSRC)/xpng_hip.hip
HIPHDR := $(wildcard $(CSRC)/*.hpp) include/xpng_hip.h

all: hip probes host oracle

HIPFLAGS := -O3 --offload-arch=$(ARCH) -std=c++17 -shared -fPIC -Wall -Wno-unused-function -pthread
hip: $(LIB)/libxpng_hip.so
$(LIB)/libxpng_hip.so: $(HIPSRC) $(HIPHDR)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) $(HIPSRC) -o $@
# the same source with the timing-study switches compiled in (kernel knock-outs, LDS pads, stamps, wave probe, fake devices):
# what tools/ load, never what the product or bench.py loads
probes: $(LIB)/libxpng_hip_probes.so
$(LIB)/libxpng_hip_probes.so: $(HIPSRC) $(HIPHDR)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -DXPNG_PROBES $(HIPSRC) -o $@

host: $(LIB)/libxpng.so $(BIN)/xpng $(BIN)/seven $(BIN)/tool
$(LIB)/libxpng.so: $(CSRC)/host/xpng_api.c $(CSRC)/host/seven.c include/xpng.h include/xpng_hip.h $(LIB)/libxpng_hip.so
	$(CC) -O2 -std=gnu11 -Wall -Wextra -shared -fPIC $(CSRC)/host/xpng_api.c $(CSRC)/host/seven.c -o $@ \
	    -L$(LIB) -lxpng_hip -Wl,-rpath,'$$ORIGIN'
$(BIN)/tool: $(CSRC)/host/tool_cli.c $(LIB)/libxpng.so
	@mkdir -p $(BIN)
	$(CC) -O2 -std=gnu11 -Wall -Wextra $(CSRC)/host/tool_cli.c -o $@ -L$(LIB) -lxpng -lxpng_hip -Wl,-rpath,'$$ORIGIN/../lib'
$(BIN)/seven: $(CSRC)/host/seven_cli.c $(LIB)/libxpng.so
	@mkdir -p $(BIN)
	$(CC) -O2 -std=gnu11 -Wall -Wextra $(CSRC)/host/seven_cli.c -o $@ -L$(LIB) -lxpng -lxpng_hip -ldl -Wl,-rpath,'$$ORIGIN/../lib'
$(BIN)/xpng: $(CSRC)/host/xpng_cli.c $(LIB)/libxpng.so
	@mkdir -p $(BIN)
	$(CC) -O2 -std=gnu11 -Wall -Wextra $(CSRC)/host/xpng_cli.c -o $@ -L$(LIB) -lxpng -lxpng_hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(LIB) $(BIN)
	$(MAKE) -C oracle clean
.PHONY: all hip probes host oracle clean
