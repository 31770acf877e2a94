// Counted waits of the training forward's hand-issued input loads (mlp_fwd.hip): the number of vector-memory
// operations a wave is known to issue between those loads and their `s_waitcnt vmcnt(N)`.  mlp_fwd.hip checks
// both values against the layer tables with static_asserts; audit_asm_loads.py reads them from this file to
// know which waits end a load's window in the disassembly.
#pragma once
#define NERF_INPUTS_WAIT_VIEW 22     /* L9 + L10: 3 chunk entries x 4 copies, 8 unit stores, 2 mask stores */
#define NERF_INPUTS_WAIT_NOVIEW 36   /* L7 + output layer: 4 chunk entries x 4 copies, 16 unit stores, 4 mask stores */
