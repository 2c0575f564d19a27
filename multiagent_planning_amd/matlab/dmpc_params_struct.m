function prm = dmpc_params_struct(variant, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term, tol)
% Packs the reference's positional constants into the struct dmpc_mex expects (dmpc_params in
% include/dmpc_hip.h). variant: 0 bound, 1 bound2, 2 all3, 3 hard, 4 ondemand, 5 ellip, 6 softall, 7 repair,
% 11 softall_c (solveSoftDMPC_c), 12 scp (solveDMPC: `tol` is its stopping tolerance)
if nargin < 13, tol = 0; end
prm = struct('K',K,'variant',variant,'order',order,'h',h,'rmin',rmin,'c',1/E1(3,3),'alim',alim, ...
             'Q1',Q1,'S1',S1,'term',term,'pmin',pmin(:)','pmax',pmax(:)','tol',tol);
end
