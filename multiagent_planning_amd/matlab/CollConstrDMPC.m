function [Ain_total, bin_total] = CollConstrDMPC(p,po,vo,n,k,l,Ain,r_min,A_initp)
% Shadows dmpc/matlab/CollConstrDMPC.m (same signature): one spherical row per other agent at horizon step k, linearised about p.
Ain_total = []; bin_total = [];
if isempty(l), return; end
N = size(l,3);
sel = setdiff(0:N-1, n-1);
if isempty(sel), return; end
a0 = A_initp(3*(k-1)+1:3*k,:)*[po(:); vo(:)];
prm = dmpc_params_struct(0, 0.2, size(l,2), r_min, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);   % context only
[Ain_total, bin_total] = dmpc_mex('coll_rows', prm, l, sel, k-1, k-1, p(:), a0, r_min, 1, Ain);
end
