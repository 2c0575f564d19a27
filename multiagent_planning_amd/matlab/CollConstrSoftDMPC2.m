function [Ain_total, bin_total, prev_dist] = CollConstrSoftDMPC2(p,po,vo,n,k,l,rmin,Ain,A_initp,E1,E2,order,violation)
% Shadows dmpc/matlab/CollConstrSoftDMPC2.m (same signature): as CollConstrSoftDMPC, rows constrain step k_ctr = k-1 (:8).
assert(order == 2, 'only order = 2 is supported');
sel = find(violation(:)') - 1; sel = sel(sel ~= n-1);
kc = k - 1;
a0 = A_initp(3*(kc-1)+1:3*kc,:)*[po(:); vo(:)];
prm = dmpc_params_struct(1, 0.2, size(l,2), rmin, [-1 -1 0], [1 1 1], 1, 1000, 100, E1, order, -5e4);   % context only
[Ain_total, bin_total, prev_dist] = dmpc_mex('coll_rows', prm, l, sel, k-1, kc-1, p(:), a0, rmin, 1/E1(3,3), Ain);
end
