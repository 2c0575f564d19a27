function [p,v,a,feasible,outbound,coll] = solveSoftDMPCbound(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,A_p,A_v,Delta,Q1,S1,E1,E2,order,term)
% Drop-in replacement of dmpc/matlab/solveSoftDMPCbound.m (same signature, same return conventions)
% that runs the per-agent QP on the GPU through dmpc_mex / libdmpc_hip.so.
% The model matrices A, A_initp, A_p, A_v, Delta are accepted for signature compatibility; the
% library uses its own (bit-identical) precomputed structure.
prm = dmpc_params_struct(0, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term);
[p,v,a,st] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
coll = double(bitand(st,4) ~= 0);
outbound = double(bitand(st,2) ~= 0);
feasible = double(bitand(st,1) ~= 0 || coll);      % solveSoftDMPCbound.m:25-31: coll returns feasible = 1
if bitand(st,48), error('dmpc:capacity','internal capacity/iteration limit hit (status %d)', st); end
if ~bitand(st,1), p = []; v = []; a = []; end       % failure: empty outputs
end
