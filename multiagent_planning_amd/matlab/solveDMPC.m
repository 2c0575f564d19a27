function [p,v,a,success] = solveDMPC(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,Delta,tol,Q1,S1)
% Drop-in replacement of dmpc/matlab/solveDMPC.m over dmpc_mex: the whole SCP loop (up to k_hor passes) runs inside one kernel launch.
prm = dmpc_params_struct(12, h, K, rmin, pmin, pmax, alim, Q1, S1, eye(3), 2, -5e4, tol);
[p,v,a,st] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
success = double(bitand(st,1) ~= 0);
if ~success, p = []; v = []; a = []; end   % solveDMPC.m:58-63
end
