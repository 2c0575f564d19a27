function [p,v,a,success,outbound,coll] = solveHardDMPCOnDemand(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,A_p,A_v,Delta,Q1,S1,E1,E2,order)
% Drop-in replacement of dmpc/matlab/solveHardDMPCOnDemand.m (same signature / conventions) over dmpc_mex.
prm = dmpc_params_struct(4, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -5e4);
[p,v,a,st] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
coll = double(bitand(st,4) ~= 0);
outbound = double(bitand(st,2) ~= 0);
success = double(bitand(st,1) ~= 0 && ~outbound);
if bitand(st,48), error('dmpc:capacity','internal capacity/iteration limit hit (status %d)', st); end
if ~bitand(st,1), p = []; v = []; a = []; end
end
